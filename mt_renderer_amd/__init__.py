"""mt_renderer_amd -- MI355X-native rModel draw path (HIP) behind a C ABI (include/mtr.h).

`scene` holds host-side synthetic rModel / rTexture data generators; `api` is the ctypes binding
of libmtr.so mirroring the reference's Model / Texture interface (src/model.rs, src/texture.rs).
Importing `api` fails loudly when the HIP library has not been built: there is no CPU fallback.
"""
from . import scene  # noqa: F401
