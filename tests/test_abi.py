"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/zkt.h declares, struct sizes match, and compute entry points fail loudly
(ZKT_ERR_DEVICE) rather than fall back when there is no GPU.  No compute calls here."""
import ctypes, importlib, os
import numpy as np
import pytest
from zkt_testlib import *

zk = importlib.import_module("zk-toolkit_amd")


def test_library_exports_every_declared_symbol():
    L = zk.lib()
    names = zk.exported_symbols()
    assert len(names) >= 55
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_struct_sizes_match_header():
    # zkt_g1_affine 104 B, zkt_g2_affine 200 B, zkt_secp_affine 72 B (include/zkt.h)
    assert G1W * 8 == 104 and G2W * 8 == 200 and 9 * 8 == 72 and FQ12 * 8 == 576


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = zk.lib()
    assert L.zkt_init(-1) == ZKT_ERR_DEVICE
    a = np.zeros((1, 6), dtype=np.uint64)
    assert L.zkt_fq_mul_batch(ptr(a), ptr(a), ptr(a), 1) == ZKT_ERR_DEVICE
    with pytest.raises(zk.ZktError):
        zk.init()


def test_gt_eq_is_plain_memory_equality():
    L = zk.lib()
    a = np.arange(72, dtype=np.uint64); b = a.copy()
    assert L.zkt_gt_eq(ptr(a), ptr(b)) == 1
    b[5] ^= 1
    assert L.zkt_gt_eq(ptr(a), ptr(b)) == 0


def test_comm_shard_range_rejects_bad_ranks_and_library_needs_no_rccl_at_load_time():
    """zkt_comm_shard_range is pure host arithmetic: world < 1 or a rank outside [0, world) yields the empty range (it used to divide by zero);
    and RCCL is opened on first use, so a single-GPU consumer of libzkt_hip.so does not link it."""
    import subprocess
    L = zk.lib()
    lo, hi = ctypes.c_size_t(7), ctypes.c_size_t(7)
    for rank, world in ((0, 0), (0, -3), (-1, 4), (4, 4), (9, 2)):
        lo.value = hi.value = 7
        L.zkt_comm_shard_range(ctypes.c_size_t(1000), rank, world, ctypes.byref(lo), ctypes.byref(hi))
        assert (lo.value, hi.value) == (0, 0), (rank, world)
    L.zkt_comm_shard_range(ctypes.c_size_t(10), 2, 3, ctypes.byref(lo), ctypes.byref(hi))
    assert (lo.value, hi.value) == (7, 10)
    L.zkt_comm_shard_range(ctypes.c_size_t(10), 0, 3, None, None)             # NULL outputs are allowed
    needed = subprocess.check_output(["readelf", "-d", zk.LIB_PATH], text=True)
    assert "librccl" not in needed and "libamdhip64" in needed
    assert L.zkt_comm_rank() == -1 and L.zkt_comm_world() == 0                 # before zkt_comm_init
    L.zkt_comm_finalize()                                                       # harmless when not initialised


def test_header_is_plain_c99(tmp_path):
    """include/zkt.h is the drop-in boundary a cgo / Rust-FFI binding consumes: it must compile as C (no C++, no torch types),
    and so must the plain-C consumer that the GPU suite runs (tests/c/abi_consumer.c)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "c", "abi_consumer.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", src])


def test_generated_constants_are_in_sync(tmp_path):
    """csrc/zkt_constants.h is generated (tools/gen_constants.py from oracle/fast_model.py): the committed header must be what the generator writes"""
    import subprocess, sys
    out = tmp_path / "zkt_constants.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_constants.py"), str(out)], timeout=300)
    assert out.read_text() == open(os.path.join(ROOT, "zk-toolkit_amd", "csrc", "zkt_constants.h")).read()


def test_no_base_pointer_hazard_in_the_device_code():
    """tools/check_base_pointer.py: no device function that keeps a base pointer (s34) calls one that uses s34 as a scratch register — this toolchain lets a callee
    do that, and the caller then returns with a garbage stack pointer (DESIGN.md §5 "A compiler limit"; found twice as silent aborts on the GPU)."""
    import subprocess, sys
    so = os.path.join(ROOT, "zk-toolkit_amd", "libzkt_hip.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_base_pointer.py"), so], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "HAZARD" not in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("0 hazards") >= 8
