"""Direct RCCL all-gather through ctypes, for hosts whose per-frame budget is tens of microseconds: a
torch.distributed collective costs ~20 us of host time per call, ncclAllGather ~3 us.  The communicator is created from
a unique id that rank 0 generates and the caller distributes (e.g. torch.distributed.broadcast_object_list).
The library is the librccl.so PyTorch already loaded (same one torch.distributed's "nccl" backend uses)."""
from __future__ import annotations

import ctypes as C
import glob
import os
from typing import Optional

NCCL_UNIQUE_ID_BYTES = 128
ncclUint8 = 1  # ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


def _find_lib() -> Optional[str]:
    try:
        import torch
        cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
        if cands:
            return sorted(cands)[0]
    except ImportError:
        pass
    for p in ("/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"):
        if os.path.exists(p):
            return p
    return None


class Rccl:
    def __init__(self):
        path = _find_lib()
        if not path:
            raise OSError("librccl.so not found")
        self.lib = C.CDLL(path)
        self.lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        self.lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
        self.lib.ncclGetErrorString.restype = C.c_char_p
        self.lib.ncclGetErrorString.argtypes = [C.c_int]
        self.comm = C.c_void_p()

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(rc).decode()}")

    def unique_id(self) -> bytes:
        uid = _UniqueId()
        self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        return bytes(uid.internal) if len(bytes(uid.internal)) == NCCL_UNIQUE_ID_BYTES else C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES)

    def init(self, uid: bytes, world: int, rank: int):
        """call with the current HIP device already selected (torch.cuda.set_device)"""
        u = _UniqueId()
        C.memmove(C.byref(u), uid, NCCL_UNIQUE_ID_BYTES)
        self._check(self.lib.ncclCommInitRank(C.byref(self.comm), world, u, rank), "ncclCommInitRank")

    def all_gather_u8(self, send_devptr: int, recv_devptr: int, nbytes: int, stream: int):
        self._check(self.lib.ncclAllGather(C.c_void_p(send_devptr), C.c_void_p(recv_devptr), nbytes, ncclUint8, self.comm,
                                           C.c_void_p(stream)), "ncclAllGather")

    @property
    def allgather_addr(self) -> int:
        """address of ncclAllGather, for a native caller (mtr_device_exchange_start)"""
        return C.cast(self.lib.ncclAllGather, C.c_void_p).value

    @property
    def comm_handle(self) -> int:
        return self.comm.value or 0

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
