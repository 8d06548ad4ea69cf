"""CPU: the HOST side of libmtr.so (csrc/mtr_api.cpp + csrc/mtr_files.cpp) compiled by g++ with AddressSanitizer and
UBSan over a stand-in HIP runtime (tests/cpp/hip_stub: device memory = host heap, so every copy is bounds-checked), then
driven with thousands of models, textures, batches and frames built from mostly malformed arguments.  GPU sanitizers
are not available on the GPU pool; this covers what the host must validate before a kernel may trust it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_api_under_address_sanitizer(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "tests", "cpp", "hip_stub"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_fuzz.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe, "6000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-3000:])
    created, rejected = [int(t.split("=")[1]) for t in r.stdout.split()]
    assert created > 1500 and rejected > 1500, r.stdout


def test_exchange_thread_under_thread_sanitizer(tmp_path):
    """render thread + the device's exchange thread over the stub runtime: 4000 frames begun, drawn, handed over, packed,
    "gathered", unpacked and destroyed; ThreadSanitizer reports any unsynchronised access to shared host state (removing
    the pool mutex from mtr_frame_destroy makes it fire)."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "exchange_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-w",
                           "-I", os.path.join(ROOT, "tests", "cpp", "hip_stub"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "exchange_tsan.cpp"), "-o", exe, "-lz", "-pthread"])
    r = subprocess.run([exe, "4000"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.returncode, r.stdout[-300:], r.stderr[-3000:])
    assert int(r.stdout.strip().split("=")[1]) > 2500


def test_no_rank_skips_a_collective_under_thread_sanitizer(tmp_path):
    """two ranks (two devices, an exchange thread each) joined by an all-gather that completes only when both have issued it;
    rank 1's pack fails at one frame, "overflowed" frames are re-run by the exchange threads while the render threads flip
    parts_disp and the palette of the model they draw.  No hang, as many all-gathers as frames on both ranks, the error
    comes out of rank 1's drain (and of the MIN the host takes over the ranks), no data race (tests/cpp/exchange_ranks_tsan.cpp)."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "exchange_ranks_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-w",
                           "-I", os.path.join(ROOT, "tests", "cpp", "hip_stub"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "exchange_ranks_tsan.cpp"), "-o", exe, "-lz", "-pthread"])
    r = subprocess.run([exe, "900"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.returncode, r.stdout[-300:], r.stderr[-3000:])
    assert "handed=1800" in r.stdout and "agreed_status=4" in r.stdout and 'rank1_error="reported"' in r.stdout, r.stdout


def test_group_gather_under_address_sanitizer(tmp_path):
    """mtr_group_* (csrc/mtr_group.cpp) over the stub runtime: groups of 1-5 ranks, odd frame sizes, every ownership map,
    explicit bands with empty ranks, parts that "overflow" and are re-run; the stub pack / unpack move a byte derived from
    the bin id, so the gathered image is right only if every shard landed at its rank's offset (tests/cpp/group_asan.cpp)."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "group_asan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-w",
                           "-I", os.path.join(ROOT, "tests", "cpp", "hip_stub"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "group_asan.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe, "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-3000:])
    out = dict(t.split("=") for t in r.stdout.split())
    assert int(out["frames"]) == 360 and int(out["reruns"]) > 20, r.stdout
