"""One-off wider sweep of the randomised GPU tests (more seeds than the suite runs, plus forced remainder-kernel and
all-frames-culling variants): hunts for rare parity failures.   python tools/probe/stress_random.py [nseeds]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mt_renderer_amd import api
import tests.test_gpu_sharding as ts
import tests.test_gpu_states as tst
import tests.test_gpu_clipping as tc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = api.Device(0)
os.environ["MTR_GEOM_SLOTS"] = "1"  # read once, at device creation: this device sends nearly every kept instance through k_geom_rest
dev_rest = api.Device(0)
os.environ.pop("MTR_GEOM_SLOTS")
fails = 0
def run(name, fn, *a, device=None):
    global fails
    try:
        fn(device or dev, *a)
    except Exception:
        fails += 1
        print("FAIL", name, a, flush=True)
        traceback.print_exc()
for seed in range(6, 6 + n):
    for d in (dev, dev_rest):
        run("hostile culling", ts.test_culling_is_conservative_on_hostile_inputs, seed, device=d)
    if seed % 8 == 0: print("seed", seed, "fails so far", fails, flush=True)
for seed in range(8, 8 + n):
    run("state mix", tst.test_random_state_mixes_match_the_oracle, seed)
    for mode in (api.GEOM_CULL_ALL_FRAMES, api.GEOM_CULL_SHARDED):
        dev.set_culling(mode)
    dev.set_culling(api.GEOM_CULL_ALL_FRAMES)
    run("state mix, all-frames culling", tst.test_random_state_mixes_match_the_oracle, seed)
    dev.set_culling(api.GEOM_CULL_SHARDED)
clip_tests = [getattr(tc, k) for k in dir(tc) if k.startswith("test_")]
print("clipping tests available:", [f.__name__ for f in clip_tests])
print("done: fails", fails)
