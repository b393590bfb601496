"""Host-side loader for the MI355X-native engine (ctypes over the C ABI in include/zkt.h).

Import with  importlib.import_module("zk-toolkit_amd")  (the directory name carries a hyphen).
The HIP library is the only compute path: if libzkt_hip.so is missing or no GPU is present,
`lib()` / `init()` raise — there is no CPU fallback."""
import ctypes, os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # read by the HIP runtime at first use: stage streams on distinct HW queues

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKT_LIB_PATH", os.path.join(_HERE, "libzkt_hip.so"))   # override only for A/B experiments
HEADER = os.path.join(os.path.dirname(_HERE), "include", "zkt.h")

ZKT_OK, ZKT_ERR_INV_ZERO, ZKT_ERR_INFINITY, ZKT_ERR_SHAPE, ZKT_ERR_DEVICE = 0, 1, 2, 3, 4
G1_WORDS64, G2_WORDS64, FQ12_WORDS64 = 13, 25, 72
G1_PARTIAL_WORDS, G2_PARTIAL_WORDS, SECP_PARTIAL_WORDS = 42, 84, 24
GROTH16_PARTIAL_WORDS = 2 * G1_PARTIAL_WORDS + G2_PARTIAL_WORDS   # ZKT_*_PARTIAL_WORDS: u32 words of one opaque Jacobian partial

_lib = None


class ZktError(RuntimeError):
    def __init__(self, status, index=None):
        self.status, self.index = status, index
        msg = lib().zkt_strerror(status).decode()
        super().__init__(f"zkt status {status}: {msg}" + (f" (element {index})" if index is not None else ""))


def lib():
    """ctypes handle on libzkt_hip.so; raises if the extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP library is the only compute path, there is no CPU fallback)")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7.  Importing torch first makes
        # this library's DT_NEEDED libamdhip64.so.7 resolve to the copy torch already mapped, so device
        # pointers and streams are shared.  (A C/C++/Rust host without torch simply uses /opt/rocm's.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        L.zkt_strerror.restype = ctypes.c_char_p
        L.zkt_last_error_index.restype = ctypes.c_size_t
        L.zkt_last_kernel_ms.restype = ctypes.c_float
        L.zkt_last_kernel_name.restype = ctypes.c_char_p
        L.zkt_g1_bases_len.restype = ctypes.c_size_t
        L.zkt_g1_msm_workspace_bytes.restype = ctypes.c_size_t
        L.zkt_g1_msm_workspace_bytes.argtypes = [ctypes.c_size_t]
        # size_t arguments beyond the sixth travel on the stack: declare them, or ctypes pushes a 32-bit int with garbage above it
        vp = ctypes.c_void_p
        L.zkt_groth16_verify_batch.argtypes = [vp, vp, vp, vp, vp, ctypes.c_size_t, ctypes.c_size_t, vp]
        L.zkt_groth16_prove.argtypes = [vp, vp, vp, vp, vp, ctypes.c_size_t, vp, vp, vp, vp, vp]
        L.zkt_groth16_setup_r1cs.argtypes = [ctypes.c_size_t] * 3 + [vp] * 10
        L.zkt_groth16_prove_r1cs.argtypes = [vp] * 7
        L.zkt_groth16_prove_r1cs_dev.argtypes = [vp] * 7
        L.zkt_groth16_setup_r1cs_sharded.argtypes = [ctypes.c_size_t] * 3 + [vp] * 8 + [ctypes.c_size_t] * 2 + [vp] * 2
        L.zkt_groth16_prove_r1cs_partials.argtypes = [vp] * 5
        L.zkt_groth16_prove_r1cs_submit.argtypes = [vp, ctypes.c_int, vp, vp, vp]
        L.zkt_groth16_prove_r1cs_collect.argtypes = [vp, ctypes.c_int, vp, vp, vp]
        L.zkt_groth16_pk_free.argtypes = [vp]; L.zkt_groth16_pk_free.restype = None
        sz = ctypes.c_size_t
        L.zkt_pairing_product_check_batch.argtypes = [vp, vp, vp, sz, sz, vp]
        L.zkt_pinocchio_prove.argtypes = [vp, vp, vp, sz, vp, vp, vp]
        L.zkt_bls_hash_to_g2_batch.argtypes = [vp, vp, sz, vp]
        L.zkt_bls_sign_batch.argtypes = [vp, vp, vp, sz, vp]
        L.zkt_bls_public_keys_batch.argtypes = [vp, sz, vp]
        L.zkt_bls_verify_batch.argtypes = [vp, vp, vp, vp, sz, vp]
        for grp in ("g1", "g2", "secp"):
            getattr(L, f"zkt_{grp}_msm_submit").argtypes = [vp, vp, sz, vp, ctypes.c_int]
            getattr(L, f"zkt_{grp}_msm_collect").argtypes = [vp, ctypes.c_int, vp, vp]
        for f in ("fq", "fr", "sp", "sn"):
            getattr(L, f"zkt_{f}_pow_batch").argtypes = [vp, vp, sz, ctypes.c_int, vp, sz]
            getattr(L, f"zkt_{f}_pow_seq").argtypes = [vp, sz, vp]
            getattr(L, f"zkt_{f}_repeat").argtypes = [vp, sz, vp]
        for grp in ("g1", "g2", "secp"):
            getattr(L, f"zkt_{grp}_is_on_curve_batch").argtypes = [vp, vp, sz]
            getattr(L, f"zkt_{grp}_in_subgroup_batch").argtypes = [vp, vp, sz]
            getattr(L, f"zkt_{grp}_generator").argtypes = [vp]; getattr(L, f"zkt_{grp}_generator").restype = None
        # multi-GPU entry points (csrc/zkt_comm.cpp)
        L.zkt_comm_init.argtypes = [ctypes.c_int, ctypes.c_int, vp]
        L.zkt_comm_init_callback.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp]
        L.zkt_comm_finalize.restype = None
        L.zkt_comm_shard_range.argtypes = [sz, ctypes.c_int, ctypes.c_int, vp, vp]; L.zkt_comm_shard_range.restype = None
        for grp in ("g1", "g2", "secp"):
            getattr(L, f"zkt_{grp}_msm_sharded").argtypes = [vp, vp, sz, vp, vp]
            getattr(L, f"zkt_{grp}_msm_sharded_collect").argtypes = [vp, ctypes.c_int, vp]
        L.zkt_groth16_prove_r1cs_sharded.argtypes = [vp] * 7
        L.zkt_bp_ipa_ctx_create.argtypes = [sz, vp, vp, vp, vp]
        L.zkt_bp_ipa_ctx_free.argtypes = [vp]; L.zkt_bp_ipa_ctx_free.restype = None
        L.zkt_bp_inner_product_argument_ctx.argtypes = [vp] * 6
        L.zkt_bp_range_proof_ctx.argtypes = [vp] * 6 + [ctypes.c_int] + [vp] * 3
        _lib = L
    return _lib


def init(device=-1):
    rc = lib().zkt_init(int(device))
    if rc != ZKT_OK:
        raise ZktError(rc)


def check(rc):
    if rc != ZKT_OK:
        raise ZktError(rc, lib().zkt_last_error_index() if rc in (ZKT_ERR_INV_ZERO, ZKT_ERR_INFINITY) else None)


def exported_symbols():
    """Every zkt_* function declared in include/zkt.h (for the ABI completeness test)."""
    import re
    names = []
    with open(HEADER) as f:
        txt = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    for m in re.finditer(r"\b(zkt_[a-z0-9_]+)\s*\(", txt):
        if m.group(1) not in names:
            names.append(m.group(1))
    return names
