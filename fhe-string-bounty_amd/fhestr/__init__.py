"""fhestr -- ctypes binding of libfhestr.so (the C ABI in include/fhestr.h).

This is the thin host-side mirror used by tests and bench.py: it marshals numpy arrays and raw
device pointers into the C ABI and nothing else.  All compute happens in the HIP library; when the
library or a GPU is missing every call fails loudly -- there is no CPU fallback here.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FHESTR_LIB: load another build of the same library (kernel A/B experiments, scripts/ab_bench.sh)
LIB_PATH = os.environ.get("FHESTR_LIB") or os.path.join(os.path.dirname(_HERE), "libfhestr.so")


class FheError(RuntimeError):
    pass


class _Params(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("k", C.c_uint32), ("N", C.c_uint32),
        ("pbs_base_log", C.c_uint32), ("pbs_level", C.c_uint32),
        ("ks_base_log", C.c_uint32), ("ks_level", C.c_uint32),
        ("msg_mod", C.c_uint32), ("carry_mod", C.c_uint32),
        ("lwe_std", C.c_double), ("glwe_std", C.c_double),
        ("grouping_factor", C.c_uint32),
    ]


@dataclass(frozen=True)
class Params:
    """shortint ClassicPBSParameters (reference: tfhe/src/shortint/parameters/mod.rs:61-76)."""
    n: int
    k: int
    N: int
    pbs_base_log: int
    pbs_level: int
    ks_base_log: int
    ks_level: int
    msg_mod: int
    carry_mod: int
    lwe_std: float
    glwe_std: float
    name: str = ""
    grouping: int = 1          # multi-bit PBS grouping factor (1 = classic PBS)

    @property
    def n_ggsw(self) -> int:
        """GGSWs in the bootstrapping key: n (classic) or n/g * 2^g (multi-bit)."""
        return self.n if self.grouping <= 1 else self.n // self.grouping * (1 << self.grouping)

    @property
    def big_size(self) -> int:
        return self.k * self.N + 1

    @property
    def small_size(self) -> int:
        return self.n + 1

    @property
    def glwe_len(self) -> int:
        return (self.k + 1) * self.N

    @property
    def delta(self) -> int:
        return (1 << 63) // (self.msg_mod * self.carry_mod)

    @property
    def ksk_len(self) -> int:
        return self.k * self.N * self.ks_level * (self.n + 1)

    @property
    def bsk_len(self) -> int:
        return self.n_ggsw * self.pbs_level * (self.k + 1) ** 2 * self.N

    def c(self) -> _Params:
        return _Params(self.n, self.k, self.N, self.pbs_base_log, self.pbs_level, self.ks_base_log,
                       self.ks_level, self.msg_mod, self.carry_mod, self.lwe_std, self.glwe_std, self.grouping)


# reference: tfhe/src/shortint/parameters/mod.rs:703-717, :658-672, :613-627
PARAM_MESSAGE_2_CARRY_2_KS_PBS = Params(742, 1, 2048, 23, 1, 3, 5, 4, 4,
                                        0.000007069849454709433, 0.00000000000000029403601535432533,
                                        "PARAM_MESSAGE_2_CARRY_2_KS_PBS")
PARAM_MESSAGE_4_CARRY_4_KS_PBS = Params(996, 1, 32768, 15, 2, 3, 7, 16, 16,
                                        6.767666038309478e-08, 2.168404344971009e-19,
                                        "PARAM_MESSAGE_4_CARRY_4_KS_PBS")
# shortint/parameters/multi_bit.rs:115-135
PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS = Params(818, 1, 2048, 22, 1, 5, 3, 4, 4,
                                                          0.000002226459789930014, 0.0000000000000003152931493498455,
                                                          "PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS", 2)
# shortint/parameters/multi_bit.rs:173-190
PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS = Params(888, 1, 2048, 21, 1, 7, 2, 4, 4,
                                                          0.0000006125031601933181, 0.0000000000000003152931493498455,
                                                          "PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS", 3)
# shortint/parameters/multi_bit.rs:96-113, :154-171 (N = 512, k = 3) and :134-152, :192-209 (N = 8192, two levels):
# served by the two-kernel path (multibit_combine_generic_kernel + the classic kernels' EXTPROD mode)
PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_2_KS_PBS = Params(764, 3, 512, 18, 1, 6, 2, 2, 2,
                                                          0.000006025673585415336, 0.0000000000039666089171633006,
                                                          "PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_2_KS_PBS", 2)
PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_3_KS_PBS = Params(765, 3, 512, 18, 1, 6, 2, 2, 2,
                                                          0.000005915594083804978, 0.0000000000039666089171633006,
                                                          "PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_3_KS_PBS", 3)
PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_2_KS_PBS = Params(922, 1, 8192, 14, 2, 4, 4, 8, 8,
                                                          0.0000003272369292345697, 0.0000000000000000002168404344971009,
                                                          "PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_2_KS_PBS", 2)
PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_3_KS_PBS = Params(972, 1, 8192, 14, 2, 6, 3, 8, 8,
                                                          0.00000013016688349592805, 0.0000000000000000002168404344971009,
                                                          "PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_3_KS_PBS", 3)
PARAM_MESSAGE_2_CARRY_1_KS_PBS = Params(742, 2, 1024, 23, 1, 4, 3, 4, 2,
                                        0.000007069849454709433, 0.00000000000000029403601535432533,
                                        "PARAM_MESSAGE_2_CARRY_1_KS_PBS")
PARAM_MESSAGE_1_CARRY_1_KS_PBS = Params(684, 3, 512, 18, 1, 4, 3, 2, 2,
                                        0.00002043784477291318, 0.0000000000034525330484572114,
                                        "PARAM_MESSAGE_1_CARRY_1_KS_PBS")

_lib = None

EXPORTS = [
    "fhe_last_error", "fhe_kernel_revision", "fhe_engine_create", "fhe_engine_destroy", "fhe_engine_params",
    "fhe_engine_load_keys", "fhe_engine_generate_keys", "fhe_engine_stream", "fhe_engine_synchronize", "fhe_engine_set_variant",
    "fhe_engine_set_multibit_combine_max", "fhe_engine_set_cluster_mode", "fhe_engine_set_keep_busy", "fhe_engine_pipeline_input_event", "fhe_engine_cluster_info", "fhe_engine_cluster_fallbacks", "fhe_engine_load_seeded_keys", "fhe_engine_set_pipeline",
    "fhe_lut_generate", "fhe_lut_upload", "fhe_lut_download", "fhe_lut_count",
    "fhe_keyswitch_batch", "fhe_pbs_batch", "fhe_ks_pbs_batch", "fhe_ks_pbs_batch_dev", "fhe_pbs_ks_batch",
    "fhe_lwe_lincomb_batch", "fhe_last_kernel_ms", "fhe_kernel_times",
    "fhe_params_ksk_len", "fhe_params_bsk_len", "fhe_client_key_create", "fhe_client_key_destroy",
    "fhe_client_encrypt", "fhe_client_decrypt", "fhe_client_gen_server_keys", "fhe_client_secret_keys",
    "fhe_random_seed", "fhe_chacha20_block", "fhe_int_plan_create", "fhe_int_plan_create_offline",
    "fhe_wire_write_lwe_ciphertext", "fhe_wire_read_lwe_ciphertext", "fhe_wire_write_keyswitch_key",
    "fhe_wire_read_keyswitch_key", "fhe_wire_write_bootstrap_key", "fhe_wire_read_bootstrap_key",
    "fhe_wire_write_shortint_ciphertext", "fhe_wire_read_shortint_ciphertext",
    "fhe_aes128_encrypt_block", "fhe_seeded_mask_words", "fhe_seeded_decompress_keyswitch_key",
    "fhe_seeded_decompress_bootstrap_key", "fhe_seeded_split_keyswitch_key", "fhe_seeded_split_bootstrap_key",
    "fhe_wire_write_seeded_keyswitch_key", "fhe_wire_read_seeded_keyswitch_key", "fhe_wire_write_seeded_bootstrap_key",
    "fhe_wire_read_seeded_bootstrap_key", "fhe_wire_write_multi_bit_bootstrap_key", "fhe_wire_read_multi_bit_bootstrap_key",
    "fhe_wire_write_compressed_server_key", "fhe_wire_read_compressed_server_key",
    "fhe_engine_expand_seeded_lwe", "fhe_seeded_decompress_lwe_batch", "fhe_wire_write_compressed_ciphertext",
    "fhe_wire_read_compressed_ciphertext", "fhe_wire_write_radix_ciphertext", "fhe_wire_read_radix_ciphertext",
    "fhe_wire_write_compressed_radix_ciphertext", "fhe_wire_read_compressed_radix_ciphertext",
    "fhe_plan_create", "fhe_plan_destroy", "fhe_plan_input", "fhe_plan_lut", "fhe_plan_lin", "fhe_plan_pbs",
    "fhe_plan_output", "fhe_plan_finalize", "fhe_plan_info", "fhe_plan_level_info", "fhe_plan_export_level",
    "fhe_plan_run", "fhe_plan_run_level_rank_dev", "fhe_plan_gather_outputs_dev", "fhe_str_plan_create",
    "fhe_plan_level_rank_info", "fhe_plan_noise_info", "fhe_noise_model", "fhe_noise_model_is_calibrated", "fhe_params_supported", "fhe_plan_run_batch", "fhe_plan_run_batch_dev", "fhe_str_op_many", "fhe_plan_set_noise_budget",
    "fhe_host_alloc", "fhe_host_free",
    "fhe_plan_pbs_signed", "fhe_plan_pbs_full_box", "fhe_plan_set_owner_hint",
    "fhe_str_to_upper", "fhe_str_to_lower", "fhe_plan_create_offline", "fhe_str_plan_create_offline",
    "fhe_str_trim_start", "fhe_str_trim_end", "fhe_str_strip", "fhe_str_replace", "fhe_str_replace_clear",
    "fhe_plan_lut_count", "fhe_plan_export_lut", "fhe_engine_set_stream", "fhe_engine_reset_stream",
    "fhe_str_len", "fhe_str_is_empty", "fhe_str_strip_prefix_clear", "fhe_str_strip_suffix_clear",
    "fhe_str_strip_prefix", "fhe_str_strip_suffix", "fhe_str_replace_general", "fhe_str_replace_clear_general",
] + [f"fhe_str_{n}{s}" for n in ("eq", "ne", "starts_with", "ends_with", "contains", "find", "rfind", "eq_ignore_case", "lt", "le", "gt", "ge", "concat")
     for s in ("", "_clear")] + ["fhe_str_repeat_clear"]


def lib() -> C.CDLL:
    """Load libfhestr.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FheError(f"{LIB_PATH} is missing: run `make -C fhe-string-bounty_amd` "
                       "(or __graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64: two HIP runtimes in
    # one process do not see the same device (the second one finds no GPU) and cannot share streams.
    # Importing torch first makes its copy the process-wide one; libfhestr.so (NEEDED libamdhip64.so.7)
    # then binds to it, so torch streams / tensors and the engine live in a single runtime.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    PP = C.POINTER(_Params)
    L.fhe_last_error.restype = C.c_char_p
    L.fhe_last_error.argtypes = []
    L.fhe_kernel_revision.restype = C.c_char_p
    L.fhe_kernel_revision.argtypes = []
    L.fhe_engine_stream.restype = vp
    L.fhe_engine_stream.argtypes = [vp]

    def sig(name, *args):
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = list(args)

    sig("fhe_engine_create", PP, i32, C.POINTER(vp))
    sig("fhe_engine_destroy", vp)
    sig("fhe_engine_params", vp, PP)
    sig("fhe_engine_load_keys", vp, vp, vp)
    sig("fhe_engine_generate_keys", vp, vp, vp, vp, vp, vp)
    sig("fhe_random_seed", vp)
    sig("fhe_chacha20_block", vp, C.c_uint64, C.c_uint64, vp)
    sig("fhe_engine_synchronize", vp)
    sig("fhe_engine_set_variant", vp, i32)
    sig("fhe_engine_set_multibit_combine_max", vp, u32)
    sig("fhe_engine_set_pipeline", vp, i32)
    sig("fhe_engine_set_cluster_mode", vp, i32, u32)
    sig("fhe_engine_set_keep_busy", vp, C.c_int)
    sig("fhe_engine_cluster_info", vp, C.POINTER(u32))
    sig("fhe_engine_cluster_fallbacks", vp, C.POINTER(u32))
    sig("fhe_engine_load_seeded_keys", vp, vp, vp, vp, vp, vp, vp)
    sig("fhe_engine_expand_seeded_lwe", vp, vp, vp, u32, vp, vp)
    sig("fhe_lut_generate", vp, vp, C.POINTER(u32), C.POINTER(C.c_uint64))
    sig("fhe_lut_upload", vp, vp, C.POINTER(u32))
    sig("fhe_lut_download", vp, u32, vp)
    sig("fhe_lut_count", vp, C.POINTER(u32))
    sig("fhe_keyswitch_batch", vp, vp, vp, u32)
    sig("fhe_pbs_batch", vp, vp, vp, vp, u32)
    sig("fhe_ks_pbs_batch", vp, vp, vp, vp, u32)
    sig("fhe_ks_pbs_batch_dev", vp, vp, vp, vp, u32)
    sig("fhe_pbs_ks_batch", vp, vp, vp, vp, u32)
    sig("fhe_lwe_lincomb_batch", vp, vp, u32, vp, vp, vp, vp, vp, u32)
    sig("fhe_last_kernel_ms", vp, C.POINTER(C.c_float))
    sig("fhe_kernel_times", vp, C.POINTER(C.c_double), C.POINTER(u32), i32)
    sig("fhe_client_key_create", PP, vp, C.POINTER(vp))
    sig("fhe_client_key_destroy", vp)
    sig("fhe_client_encrypt", vp, vp, u32, vp)
    sig("fhe_client_decrypt", vp, vp, u32, vp)
    sig("fhe_client_gen_server_keys", vp, vp, vp, i32)
    sig("fhe_client_secret_keys", vp, vp, vp)
    sig("fhe_plan_create", vp, C.POINTER(vp))
    sig("fhe_plan_destroy", vp)
    sig("fhe_plan_create_offline", PP, C.POINTER(vp))
    sig("fhe_str_plan_create_offline", PP, C.c_char_p, u32, u32, vp, u32, u32, C.POINTER(vp))
    sig("fhe_plan_lut_count", vp, C.POINTER(u32))
    sig("fhe_plan_export_lut", vp, u32, vp)
    sig("fhe_engine_set_stream", vp, vp)
    sig("fhe_engine_reset_stream", vp)
    sig("fhe_plan_input", vp, C.c_uint64, C.POINTER(u32))
    sig("fhe_plan_lut", vp, vp, C.POINTER(u32))
    sig("fhe_plan_lin", vp, vp, vp, u32, C.c_int64, C.POINTER(u32))
    sig("fhe_plan_pbs", vp, u32, u32, C.POINTER(u32))
    sig("fhe_plan_output", vp, u32)
    sig("fhe_plan_finalize", vp, u32)
    sig("fhe_plan_info", vp, C.POINTER(u32))
    sig("fhe_plan_level_info", vp, u32, C.POINTER(u32))
    sig("fhe_plan_export_level", vp, u32, vp, vp, vp, vp, vp)
    sig("fhe_plan_run", vp, vp, vp)
    sig("fhe_plan_run_batch", vp, u32, vp, vp)
    sig("fhe_plan_run_batch_dev", vp, u32, vp, vp)
    sig("fhe_str_op_many", vp, C.c_char_p, vp, u32, u32, vp, u32, vp, u32, vp, C.POINTER(u32))
    sig("fhe_plan_run_level_rank_dev", vp, vp, u32, u32)
    sig("fhe_plan_level_rank_info", vp, u32, u32, C.POINTER(u32))
    sig("fhe_plan_noise_info", vp, C.POINTER(C.c_double))
    sig("fhe_noise_model", PP, C.POINTER(C.c_double))
    sig("fhe_noise_model_is_calibrated", PP)
    sig("fhe_params_supported", PP)
    sig("fhe_host_alloc", C.c_size_t, C.POINTER(vp))
    sig("fhe_host_free", vp)
    sig("fhe_plan_set_noise_budget", vp, C.c_double)
    sig("fhe_plan_pbs_signed", vp, u32, u32, C.POINTER(u32))
    sig("fhe_plan_pbs_full_box", vp, u32, C.c_int, C.POINTER(u32))
    sig("fhe_plan_set_owner_hint", vp, i32)
    sig("fhe_plan_gather_outputs_dev", vp, vp, vp)
    sig("fhe_str_plan_create", vp, C.c_char_p, u32, u32, vp, u32, u32, C.POINTER(vp))
    sig("fhe_int_plan_create", vp, C.c_char_p, u32, C.c_uint64, u32, C.POINTER(vp))
    sig("fhe_int_plan_create_offline", PP, C.c_char_p, u32, C.c_uint64, u32, C.POINTER(vp))
    sig("fhe_str_len", vp, vp, u32, vp)
    sig("fhe_str_is_empty", vp, vp, u32, vp)
    sig("fhe_str_strip_prefix_clear", vp, vp, u32, vp, u32, vp)
    sig("fhe_str_strip_suffix_clear", vp, vp, u32, vp, u32, vp)
    sig("fhe_str_repeat_clear", vp, vp, u32, u32, vp)
    sig("fhe_str_strip_prefix", vp, vp, u32, vp, u32, vp)
    sig("fhe_str_strip_suffix", vp, vp, u32, vp, u32, vp)
    sig("fhe_str_replace_general", vp, vp, u32, vp, u32, vp, u32, u32, vp)
    sig("fhe_str_replace_clear_general", vp, vp, u32, vp, u32, vp, u32, u32, vp)
    for n in ("eq", "ne", "starts_with", "ends_with", "contains", "find", "rfind", "eq_ignore_case", "lt", "le", "gt", "ge", "concat"):
        sig(f"fhe_str_{n}", vp, vp, u32, vp, u32, vp)
        sig(f"fhe_str_{n}_clear", vp, vp, u32, vp, u32, vp)
    for n in ("trim_start", "trim_end", "strip"):
        sig(f"fhe_str_{n}", vp, vp, u32, vp)
    sig("fhe_str_replace", vp, vp, u32, vp, u32, vp)
    sig("fhe_str_replace_clear", vp, vp, u32, vp, vp, u32, vp)
    sig("fhe_str_to_upper", vp, vp, u32, vp)
    sig("fhe_str_to_lower", vp, vp, u32, vp)
    for name in ("fhe_params_ksk_len", "fhe_params_bsk_len"):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [PP]
    _lib = L
    return L


def seed_bytes(seed) -> bytes:
    """256-bit seed (ChaCha20 key) as 32 bytes.  Tests pass small integers (little endian, zero extended);
    production code passes random_seed()."""
    if isinstance(seed, (bytes, bytearray)):
        if len(seed) != 32:
            raise FheError("a seed is 32 bytes")
        return bytes(seed)
    return int(seed).to_bytes(32, "little")


def random_seed() -> bytes:
    """32 bytes from the OS CSPRNG (fhe_random_seed)."""
    buf = (C.c_uint8 * 32)()
    _check(lib().fhe_random_seed(buf))
    return bytes(buf)


def chacha20_block(key: bytes, counter: int, stream: int) -> np.ndarray:
    out = np.zeros(16, dtype=np.uint32)
    kb = (C.c_uint8 * 32)(*seed_bytes(key))
    _check(lib().fhe_chacha20_block(kb, C.c_uint64(counter), C.c_uint64(stream), _ptr(out)))
    return out


def pinned_empty(shape, dtype=np.uint64):
    """numpy array in page-locked host memory (fhe_host_alloc): host <-> GPU copies of it run at the full PCIe rate and
    never page-fault.  Freed when the array (and every view of it) is gone."""
    import weakref
    shape = tuple(int(d) for d in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if shape else 1
    ptr = C.c_void_p()
    _check(lib().fhe_host_alloc(max(1, n * dt.itemsize), C.byref(ptr)))
    buf = (C.c_uint8 * max(1, n * dt.itemsize)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    weakref.finalize(buf, lib().fhe_host_free, C.c_void_p(ptr.value))
    return arr


def params_supported(params: "Params"):
    """(True, "") if fhe_engine_create would accept the parameters, else (False, reason).  Needs no device."""
    if lib().fhe_params_supported(C.byref(params.c())) == 0:
        return True, ""
    return False, lib().fhe_last_error().decode()


def noise_model_is_calibrated(params: "Params") -> bool:
    """Has the model's V_pbs been checked against measured PBS output noise for this (N, k, level, grouping) shape?"""
    return bool(lib().fhe_noise_model_is_calibrated(C.byref(params.c())))


def noise_model(params: "Params") -> dict:
    """Variance model of one KS -> PBS (csrc/noise_model.h): variances with the torus = 1."""
    a = (C.c_double * 6)()
    _check(lib().fhe_noise_model(C.byref(params.c()), a))
    return dict(zip(("v_pbs", "v_ks", "v_ms", "half_box", "budget", "log2_pfail_at_budget"), map(float, a)))


def kernel_revision() -> str:
    return lib().fhe_kernel_revision().decode()


def _check(rc: int):
    if rc != 0:
        raise FheError(lib().fhe_last_error().decode() or "fhestr call failed")


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One GPU's evaluation engine: resident keys + LUTs, batched KS/PBS (mirrors the evaluation
    half of shortint::ServerKey, tfhe/src/shortint/server_key/mod.rs)."""

    def __init__(self, params: Params, device: int = 0, log2_points: int = 0):
        self.params = params
        self._h = C.c_void_p()
        _check(lib().fhe_engine_create(C.byref(params.c()), device, C.byref(self._h)))
        if log2_points:
            _check(lib().fhe_engine_set_variant(self._h, log2_points))

    def load_seeded_keys(self, ksk_seed, ksk_bodies, bsk_seed, bsk_bodies, export: bool = False):
        """A tfhe-rs CompressedServerKey (shortint/server_key/compressed.rs): bodies + two 128-bit compression seeds
        (16 bytes, or ints = Seed(u128)); the masks are expanded on the GPU.  export=True returns (bsk_std, ksk)."""
        p = self.params
        kb, bb = _u64(ksk_bodies), _u64(bsk_bodies)
        if kb.size != p.k * p.N * p.ks_level or bb.size != p.n_ggsw * p.pbs_level * (p.k + 1) * p.N:
            raise FheError("seeded key: body count does not match the parameter set")
        seeds = [(C.c_uint8 * 16).from_buffer_copy(s.to_bytes(16, "little") if isinstance(s, int) else bytes(s))
                 for s in (ksk_seed, bsk_seed)]
        bsk = np.zeros(p.bsk_len, dtype=np.uint64) if export else None
        ksk = np.zeros(p.ksk_len, dtype=np.uint64) if export else None
        _check(lib().fhe_engine_load_seeded_keys(self._h, seeds[0], _ptr(kb), seeds[1], _ptr(bb),
                                                 _ptr(bsk) if export else None, _ptr(ksk) if export else None))
        return (bsk, ksk) if export else None

    def expand_seeded_lwe(self, seeds, bodies, d_out: int | None = None):
        """Compressed ciphertexts (one 16-byte compression seed and one body each; shortint CompressedCiphertext) ->
        full big-key ciphertexts, masks generated on the GPU.  d_out: device pointer to write them to (count x
        big_size words) instead of returning a host array."""
        seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8).reshape(-1, 16))
        bodies = _u64(bodies)
        if seeds.shape[0] != bodies.size:
            raise FheError("one seed per body")
        out = None if d_out else np.zeros((bodies.size, self.params.big_size), dtype=np.uint64)
        _check(lib().fhe_engine_expand_seeded_lwe(self._h, seeds.ctypes.data_as(C.c_void_p), _ptr(bodies), bodies.size,
                                                  C.c_void_p(d_out) if d_out else None, _ptr(out) if out is not None else None))
        return out

    def set_variant(self, selector: int):
        """Blind-rotation variant: log2(points per thread), + 16 for the two-LWEs-per-CU layout; 0 = automatic.  After the
        keys are loaded only variants with the same points per thread (same Fourier key layout) can be chosen."""
        _check(lib().fhe_engine_set_variant(self._h, selector))

    def set_pipeline(self, mode):
        """Throughput modes for back-to-back apply_lookup_table_dev calls (include/fhestr.h, fhe_engine_set_pipeline):
        0 / False off; 1 / True the keyswitch of call k+1 in the shadow of the blind rotation of call k; 2 whole calls
        overlapped on two streams on the two-LWEs-per-CU kernel."""
        _check(lib().fhe_engine_set_pipeline(self._h, int(mode)))

    def pipeline_input_event(self, hip_event: int):
        """The next pipelined apply_lookup_table_dev call's keyswitch waits for this hipEvent_t (e.g.
        torch.cuda.Event().cuda_event after .record()); see fhe_engine_pipeline_input_event."""
        _check(lib().fhe_engine_pipeline_input_event(self._h, C.c_void_p(hip_event)))

    def set_keep_busy(self, on: bool):
        """Small launches carry replicas on the idle CUs so the GPU keeps its clock for the next large one (include/fhestr.h)."""
        _check(lib().fhe_engine_set_keep_busy(self.handle, int(bool(on))))

    def set_cluster_mode(self, mode: int, max_batch: int = 0xFFFFFFFF):
        """N >= 16384: several CUs per LWE (-1 automatic, 0 never, 1 always; include/fhestr.h)."""
        _check(lib().fhe_engine_set_cluster_mode(self._h, mode, max_batch))

    def cluster_info(self) -> int:
        """Clusters the last cluster launch formed (synchronises)."""
        n = C.c_uint32(0)
        _check(lib().fhe_engine_cluster_info(self._h, C.byref(n)))
        return n.value

    def cluster_fallbacks(self) -> int:
        """How often a multi-CU launch gave up (compute units held by another kernel) and was re-run on the one-workgroup kernel."""
        n = C.c_uint32(0)
        _check(lib().fhe_engine_cluster_fallbacks(self._h, C.byref(n)))
        return n.value

    def set_multibit_combine_max(self, max_batch: int):
        """Multi-bit PBS: batches up to max_batch prepare their GGSWs on the whole GPU first (0 = always fused)."""
        _check(lib().fhe_engine_set_multibit_combine_max(self._h, max_batch))

    def close(self):
        if self._h:
            lib().fhe_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return int(lib().fhe_engine_stream(self._h) or 0)

    def synchronize(self):
        _check(lib().fhe_engine_synchronize(self._h))

    def set_stream(self, hip_stream: int | None):
        """Launch on a caller-owned stream handle (e.g. torch.cuda.current_stream().cuda_stream; 0 is
        HIP's default stream).  None switches back to the engine's own stream."""
        if hip_stream is None:
            _check(lib().fhe_engine_reset_stream(self._h))
        else:
            _check(lib().fhe_engine_set_stream(self._h, C.c_void_p(hip_stream)))

    def load_keys(self, bsk_std, ksk):
        p = self.params
        bsk_std, ksk = _u64(bsk_std), _u64(ksk)
        if bsk_std.size != p.bsk_len or ksk.size != p.ksk_len:
            raise FheError("key size mismatch")
        _check(lib().fhe_engine_load_keys(self._h, _ptr(bsk_std), _ptr(ksk)))

    def generate_keys(self, glwe_sk, small_sk, seed: int, export: bool = False):
        """KSK + BSK generated on the device from the secret keys (ServerKey::new,
        shortint/engine/server_side.rs:54-160) and installed; export=True also returns the
        standard-domain (bsk, ksk) it generated."""
        p = self.params
        glwe_sk, small_sk = _u64(glwe_sk), _u64(small_sk)
        if glwe_sk.size != p.k * p.N or small_sk.size != p.n:
            raise FheError("secret key size mismatch")
        bsk = np.zeros(p.bsk_len, dtype=np.uint64) if export else None
        ksk = np.zeros(p.ksk_len, dtype=np.uint64) if export else None
        sb = (C.c_uint8 * 32)(*seed_bytes(seed))
        _check(lib().fhe_engine_generate_keys(self._h, _ptr(glwe_sk), _ptr(small_sk), sb,
                                              _ptr(bsk) if export else None, _ptr(ksk) if export else None))
        return (bsk, ksk) if export else None

    # shortint/server_key/mod.rs:383-399
    def generate_lookup_table(self, f):
        p = self.params
        table = np.array([int(f(i)) for i in range(p.msg_mod * p.carry_mod)], dtype=np.uint64)
        lut_id, deg = C.c_uint32(), C.c_uint64()
        _check(lib().fhe_lut_generate(self._h, _ptr(table), C.byref(lut_id), C.byref(deg)))
        return lut_id.value, deg.value

    # shortint/server_key/bivariate_pbs.rs:71-96,125-130
    def generate_lookup_table_bivariate(self, f, factor=None):
        p = self.params
        factor = factor or p.msg_mod
        return self.generate_lookup_table(
            lambda x: f((x // factor) % p.msg_mod, (x % factor) % p.msg_mod))

    def upload_lut(self, acc) -> int:
        acc = _u64(acc)
        if acc.size != self.params.glwe_len:
            raise FheError("accumulator size mismatch")
        lut_id = C.c_uint32()
        _check(lib().fhe_lut_upload(self._h, _ptr(acc), C.byref(lut_id)))
        return lut_id.value

    def download_lut(self, lut_id: int) -> np.ndarray:
        acc = np.zeros(self.params.glwe_len, dtype=np.uint64)
        _check(lib().fhe_lut_download(self._h, lut_id, _ptr(acc)))
        return acc

    def keyswitch(self, cts) -> np.ndarray:
        p = self.params
        cts = _u64(cts).reshape(-1, p.big_size)
        out = np.zeros((cts.shape[0], p.small_size), dtype=np.uint64)
        _check(lib().fhe_keyswitch_batch(self._h, _ptr(cts), _ptr(out), cts.shape[0]))
        return out

    def _idx(self, lut_idx, count):
        if lut_idx is None:
            return None, None
        idx = np.ascontiguousarray(lut_idx, dtype=np.uint32)
        if idx.size != count:
            raise FheError("lut_idx length mismatch")
        return idx, _ptr(idx)

    def pbs(self, cts_small, lut_idx=None) -> np.ndarray:
        p = self.params
        cts_small = _u64(cts_small).reshape(-1, p.small_size)
        out = np.zeros((cts_small.shape[0], p.big_size), dtype=np.uint64)
        idx, ip = self._idx(lut_idx, cts_small.shape[0])
        _check(lib().fhe_pbs_batch(self._h, _ptr(cts_small), ip, _ptr(out), cts_small.shape[0]))
        return out

    def apply_lookup_table(self, cts, lut_idx=None) -> np.ndarray:
        """Batched KS -> PBS (shortint/server_key/mod.rs:457-476,783-857)."""
        p = self.params
        cts = _u64(cts).reshape(-1, p.big_size)
        out = np.zeros_like(cts)
        idx, ip = self._idx(lut_idx, cts.shape[0])
        _check(lib().fhe_ks_pbs_batch(self._h, _ptr(cts), ip, _ptr(out), cts.shape[0]))
        return out

    def apply_lookup_table_small_key(self, cts_small, lut_idx=None) -> np.ndarray:
        """PBS -> KS order on small-key ciphertexts (shortint/server_key/mod.rs:859-932)."""
        p = self.params
        cts_small = _u64(cts_small).reshape(-1, p.small_size)
        out = np.zeros_like(cts_small)
        idx, ip = self._idx(lut_idx, cts_small.shape[0])
        _check(lib().fhe_pbs_ks_batch(self._h, _ptr(cts_small), ip, _ptr(out), cts_small.shape[0]))
        return out

    def apply_lookup_table_dev(self, d_in: int, d_lut_idx: int | None, d_out: int, count: int):
        """Device-pointer variant, asynchronous on the engine stream."""
        _check(lib().fhe_ks_pbs_batch_dev(self._h, C.c_void_p(d_in),
                                          C.c_void_p(d_lut_idx) if d_lut_idx else None,
                                          C.c_void_p(d_out), count))

    def lincomb(self, pool, jobs):
        """jobs: list of (terms=[(src, coeff), ...], const_body)."""
        p = self.params
        pool = _u64(pool).reshape(-1, p.big_size)
        off = np.zeros(len(jobs) + 1, dtype=np.uint32)
        src, coeff, cst = [], [], []
        for j, (terms, c) in enumerate(jobs):
            for s, a in terms:
                src.append(s)
                coeff.append(a)
            off[j + 1] = len(src)
            cst.append(c & (2 ** 64 - 1))
        src = np.array(src + [0], dtype=np.uint32)
        coeff = np.array(coeff + [0], dtype=np.int32)
        cst = np.array(cst, dtype=np.uint64)
        out = np.zeros((len(jobs), p.big_size), dtype=np.uint64)
        _check(lib().fhe_lwe_lincomb_batch(self._h, _ptr(pool), pool.shape[0], _ptr(off), _ptr(src),
                                           _ptr(coeff), _ptr(cst), _ptr(out), len(jobs)))
        return out

    def last_kernel_ms(self):
        ms = (C.c_float * 2)()
        _check(lib().fhe_last_kernel_ms(self._h, ms))
        return float(ms[0]), float(ms[1])

    def kernel_times(self, reset=True):
        """(keyswitch_ms_total, blind_rotate_ms_total, calls) since the last reset (HIP events)."""
        ms = (C.c_double * 2)()
        calls = C.c_uint32()
        _check(lib().fhe_kernel_times(self._h, ms, C.byref(calls), 1 if reset else 0))
        return float(ms[0]), float(ms[1]), int(calls.value)


class ClientKey:
    """Client side (CPU): mirrors shortint::ClientKey (tfhe/src/shortint/client_key/mod.rs)."""

    def __init__(self, params: Params, seed):
        """seed: 32 bytes (random_seed()) or, for reproducible tests, an int."""
        self.params = params
        self._h = C.c_void_p()
        sb = (C.c_uint8 * 32)(*seed_bytes(seed))
        _check(lib().fhe_client_key_create(C.byref(params.c()), sb, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().fhe_client_key_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encrypt(self, msgs) -> np.ndarray:
        msgs = _u64(np.atleast_1d(msgs))
        cts = np.zeros((msgs.size, self.params.big_size), dtype=np.uint64)
        _check(lib().fhe_client_encrypt(self._h, _ptr(msgs), msgs.size, _ptr(cts)))
        return cts

    def decrypt(self, cts) -> np.ndarray:
        """message-and-carry value of every ciphertext."""
        cts = _u64(cts).reshape(-1, self.params.big_size)
        out = np.zeros(cts.shape[0], dtype=np.uint64)
        _check(lib().fhe_client_decrypt(self._h, _ptr(cts), cts.shape[0], _ptr(out)))
        return out.astype(np.int64)

    def gen_server_keys(self, threads: int | None = None):
        p = self.params
        bsk = np.zeros(p.bsk_len, dtype=np.uint64)
        ksk = np.zeros(p.ksk_len, dtype=np.uint64)
        _check(lib().fhe_client_gen_server_keys(self._h, _ptr(bsk), _ptr(ksk),
                                                threads or min(16, os.cpu_count() or 1)))
        return bsk, ksk

    def secret_keys(self):
        p = self.params
        g = np.zeros(p.k * p.N, dtype=np.uint64)
        s = np.zeros(p.n, dtype=np.uint64)
        _check(lib().fhe_client_secret_keys(self._h, _ptr(g), _ptr(s)))
        return g, s


class Plan:
    """A levelised shortint circuit (include/fhestr.h, "plans").  Build with input/lut/lin/pbs/output
    + finalize, or get a ready-made FheString operation from Plan.string_op."""

    def __init__(self, engine: "Engine | None", handle=None, params: Params | None = None):
        """engine=None builds an offline plan (exportable, not runnable) from `params`."""
        self.engine = engine
        self.params = engine.params if engine is not None else params
        self._h = handle or C.c_void_p()
        if handle is None:
            if engine is not None:
                _check(lib().fhe_plan_create(engine.handle, C.byref(self._h)))
            else:
                _check(lib().fhe_plan_create_offline(C.byref(params.c()), C.byref(self._h)))

    @classmethod
    def string_op(cls, engine: "Engine | None", op: str, a_cap: int, b_cap: int = 0,
                  clear: bytes | None = None, world: int = 1, params: Params | None = None) -> "Plan":
        h = C.c_void_p()
        buf = (C.c_uint8 * max(1, len(clear or b"")))(*(clear or b""))
        if engine is not None:
            _check(lib().fhe_str_plan_create(engine.handle, op.encode(), a_cap, b_cap, buf,
                                             len(clear or b""), world, C.byref(h)))
        else:
            _check(lib().fhe_str_plan_create_offline(C.byref(params.c()), op.encode(), a_cap, b_cap, buf,
                                                     len(clear or b""), world, C.byref(h)))
        return cls(engine, h, params)

    @classmethod
    def integer_op(cls, engine: "Engine | None", op: str, n_blocks: int, scalar: int = 0, world: int = 1,
                   params: Params | None = None) -> "Plan":
        """Radix-integer operation on n_blocks blocks (include/fhestr.h, "radix-integer operations")."""
        h = C.c_void_p()
        if engine is not None:
            _check(lib().fhe_int_plan_create(engine.handle, op.encode(), n_blocks, C.c_uint64(scalar), world, C.byref(h)))
        else:
            _check(lib().fhe_int_plan_create_offline(C.byref(params.c()), op.encode(), n_blocks, C.c_uint64(scalar),
                                                     world, C.byref(h)))
        return cls(engine, h, params)

    def export_luts(self) -> dict:
        """{plan-local LUT id: accumulator}."""
        n = C.c_uint32()
        _check(lib().fhe_plan_lut_count(self._h, C.byref(n)))
        out = {}
        for i in range(n.value):
            acc = np.zeros(self.params.glwe_len, dtype=np.uint64)
            _check(lib().fhe_plan_export_lut(self._h, i, _ptr(acc)))
            out[i] = acc
        return out

    def close(self):
        if self._h:
            lib().fhe_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- building ----
    def input(self, degree: int | None = None) -> int:
        node = C.c_uint32()
        d = self.params.msg_mod - 1 if degree is None else degree
        _check(lib().fhe_plan_input(self._h, d, C.byref(node)))
        return node.value

    def lut(self, f) -> int:
        p = self.params
        table = np.array([int(f(i)) for i in range(p.msg_mod * p.carry_mod)], dtype=np.uint64)
        out = C.c_uint32()
        _check(lib().fhe_plan_lut(self._h, _ptr(table), C.byref(out)))
        return out.value

    def lin(self, terms, constant: int = 0) -> int:
        nodes = np.array([t[0] for t in terms] + [0], dtype=np.uint32)
        coeffs = np.array([t[1] for t in terms] + [0], dtype=np.int32)
        out = C.c_uint32()
        _check(lib().fhe_plan_lin(self._h, _ptr(nodes), _ptr(coeffs), len(terms), constant, C.byref(out)))
        return out.value

    def pbs(self, src: int, lut: int, signed: bool = False) -> int:
        """apply_lookup_table; signed=True declares that the input may be negative (padding bit in use)."""
        out = C.c_uint32()
        _check((lib().fhe_plan_pbs_signed if signed else lib().fhe_plan_pbs)(self._h, src, lut, C.byref(out)))
        return out.value

    def pbs_full_box(self, src: int, all: bool) -> int:
        """msg*carry bits reduced in one lookup: (sum == msg*carry) if all else (sum != 0); see include/fhestr.h."""
        out = C.c_uint32()
        _check(lib().fhe_plan_pbs_full_box(self._h, src, int(bool(all)), C.byref(out)))
        return out.value

    def set_owner_hint(self, rank: int):
        """PBS nodes created from now on run on `rank` (-1 = automatic); see include/fhestr.h."""
        _check(lib().fhe_plan_set_owner_hint(self._h, rank))

    def set_noise_budget(self, budget: float):
        _check(lib().fhe_plan_set_noise_budget(self._h, budget))

    def output(self, node: int):
        _check(lib().fhe_plan_output(self._h, node))

    def finalize(self, world: int = 1):
        _check(lib().fhe_plan_finalize(self._h, world))

    # ---- queries ----
    def info(self) -> dict:
        a = (C.c_uint32 * 6)()
        _check(lib().fhe_plan_info(self._h, a))
        return dict(zip(("n_inputs", "n_outputs", "n_levels", "n_pbs", "pool_slots", "world"), map(int, a)))

    def noise_info(self) -> dict:
        a = (C.c_double * 4)()
        _check(lib().fhe_plan_noise_info(self._h, a))
        return dict(zip(("max_pbs_input_noise", "budget", "log2_pfail_worst", "gathered_lwes"), map(float, a)))

    def level_info(self, level: int) -> dict:
        a = (C.c_uint32 * 8)()
        _check(lib().fhe_plan_level_info(self._h, level, a))
        return dict(zip(("jobs", "local_base", "local_size", "e_max", "recv_base", "terms"), map(int, a)))

    def level_rank_info(self, level: int, rank: int) -> dict:
        a = (C.c_uint32 * 3)()
        _check(lib().fhe_plan_level_rank_info(self._h, level, rank, a))
        return dict(zip(("job_lo", "job_hi", "n_export"), map(int, a)))

    def export_level(self, level: int) -> dict:
        li = self.level_info(level)
        off = np.zeros(li["jobs"] + 1, dtype=np.uint32)
        src = np.zeros(max(li["terms"], 1), dtype=np.uint32)
        coeff = np.zeros(max(li["terms"], 1), dtype=np.int32)
        cst = np.zeros(max(li["jobs"], 1), dtype=np.uint64)
        n_lut = li["jobs"] if level < self.info()["n_levels"] else 0
        lut = np.zeros(max(n_lut, 1), dtype=np.uint32)
        _check(lib().fhe_plan_export_level(self._h, level, _ptr(off), _ptr(src), _ptr(coeff), _ptr(cst),
                                           _ptr(lut) if n_lut else None))
        return dict(li, off=off, src=src[:li["terms"]], coeff=coeff[:li["terms"]], cst=cst[:li["jobs"]],
                    lut=lut[:n_lut])

    # ---- execution ----
    def run(self, inputs) -> np.ndarray:
        """Single GPU, host buffers."""
        p = self.params
        info = self.info()
        inputs = _u64(inputs).reshape(-1, p.big_size)
        if inputs.shape[0] != info["n_inputs"]:
            raise FheError(f"plan expects {info['n_inputs']} input ciphertexts, got {inputs.shape[0]}")
        out = np.zeros((info["n_outputs"], p.big_size), dtype=np.uint64)
        _check(lib().fhe_plan_run(self._h, _ptr(inputs) if inputs.size else None, _ptr(out)))
        return out

    def run_batch(self, inputs) -> np.ndarray:
        """Many independent instances of the plan in one pass (fhe_plan_run_batch): inputs (instances, n_inputs, kN+1)
        -> outputs (instances, n_outputs, kN+1).  Level l of all instances is one launch."""
        p = self.params
        info = self.info()
        inputs = _u64(inputs)
        if inputs.ndim != 3 or inputs.shape[1:] != (info["n_inputs"], p.big_size):
            raise FheError(f"run_batch expects (instances, {info['n_inputs']}, {p.big_size}) ciphertext words, got {inputs.shape}")
        out = np.zeros((inputs.shape[0], info["n_outputs"], p.big_size), dtype=np.uint64)
        _check(lib().fhe_plan_run_batch(self._h, inputs.shape[0], _ptr(inputs) if inputs.size else None, _ptr(out)))
        return out

    def run_batch_dev(self, d_inputs: int, d_outputs: int, instances: int):
        """The same on device arrays (raw pointers), ordered on the engine's stream; no host synchronisation."""
        _check(lib().fhe_plan_run_batch_dev(self._h, instances, C.c_void_p(d_inputs), C.c_void_p(d_outputs)))

    def run_level_rank_dev(self, d_pool: int, level: int, rank: int):
        _check(lib().fhe_plan_run_level_rank_dev(self._h, C.c_void_p(d_pool), level, rank))

    def gather_outputs_dev(self, d_pool: int, d_out: int):
        _check(lib().fhe_plan_gather_outputs_dev(self._h, C.c_void_p(d_pool), C.c_void_p(d_out)))


def int_to_blocks(params: Params, value: int, n_blocks: int) -> np.ndarray:
    """Little-endian radix digits of an unsigned integer (integer/block_decomposition.rs:119-144)."""
    bits = params.msg_mod.bit_length() - 1
    return np.array([(value >> (bits * i)) & (params.msg_mod - 1) for i in range(n_blocks)], dtype=np.uint64)


def blocks_to_int(params: Params, blocks) -> int:
    bits = params.msg_mod.bit_length() - 1
    return sum((int(b) % params.msg_mod) << (bits * i) for i, b in enumerate(np.asarray(blocks).reshape(-1)))


def blocks_per_char(params: Params) -> int:
    return 8 // (params.msg_mod.bit_length() - 1)


def string_to_blocks(params: Params, s: bytes, cap: int) -> np.ndarray:
    """Zero padded, little-endian block digits of every character (integer/block_decomposition.rs:119-144)."""
    if len(s) > cap:
        raise FheError("string longer than its capacity")
    bits = params.msg_mod.bit_length() - 1
    data = np.frombuffer(s.ljust(cap, b"\0"), dtype=np.uint8).astype(np.uint64)
    return np.stack([(data >> (bits * b)) & (params.msg_mod - 1) for b in range(8 // bits)], axis=1).reshape(-1)


def blocks_to_string(params: Params, blocks) -> bytes:
    bits = params.msg_mod.bit_length() - 1
    bpc = 8 // bits
    b = (np.asarray(blocks).reshape(-1, bpc) % params.msg_mod).astype(np.uint64)
    vals = sum(b[:, i] << (bits * i) for i in range(bpc))
    return bytes(int(v) for v in vals).rstrip(b"\0")


class FheStringOps:
    """FheString operator surface over one engine (eq/ne/starts_with/ends_with/contains/find/
    to_upper/to_lower).  Strings are (cap*blocks, kN+1) arrays of big-key LWEs (see string_to_blocks)."""

    def __init__(self, engine: Engine, out_alloc=None):
        """out_alloc(shape) -> uint64 array the results are written into (default: np.zeros; pass fhestr.pinned_empty --
        or a cache of such buffers -- for large strings: pageable memory moves at a fraction of the PCIe rate)."""
        self.engine = engine
        self.bpc = blocks_per_char(engine.params)
        self._alloc = out_alloc or (lambda shape: np.zeros(shape, dtype=np.uint64))

    def _cap(self, ct):
        ct = _u64(ct).reshape(-1, self.engine.params.big_size)
        return ct, ct.shape[0] // self.bpc

    def _binary(self, op, a, b):
        a, a_cap = self._cap(a)
        n_dig = 0
        while (self.engine.params.msg_mod ** n_dig) < a_cap + 1:
            n_dig += 1
        n_out = 1 + n_dig if op in ("find", "rfind") else 1
        out = self._alloc((n_out, self.engine.params.big_size))
        if isinstance(b, (bytes, bytearray)):
            buf = (C.c_uint8 * max(1, len(b)))(*b)
            _check(getattr(lib(), f"fhe_str_{op}_clear")(self.engine.handle, _ptr(a), a_cap, buf, len(b), _ptr(out)))
        else:
            b, b_cap = self._cap(b)
            _check(getattr(lib(), f"fhe_str_{op}")(self.engine.handle, _ptr(a), a_cap, _ptr(b), b_cap, _ptr(out)))
        return out

    def op_many(self, op, rows, b=None):
        """`op` on every row against ONE second operand in a single pass (fhe_str_op_many): rows (count, cap*blocks, kN+1);
        b: an encrypted (zero padded) string, clear bytes, or None for unary operations.  Returns (count, n_outputs, kN+1)."""
        big = self.engine.params.big_size
        rows = _u64(rows)
        if rows.ndim != 3 or rows.shape[2] != big:
            raise FheError(f"op_many expects rows of shape (count, cap*blocks, {big})")
        count, a_cap = rows.shape[0], rows.shape[1] // self.bpc
        clear = b if isinstance(b, (bytes, bytearray)) else None
        enc = None if (b is None or clear is not None) else self._cap(b)
        name = op + ("_clear" if clear is not None else "")
        buf = (C.c_uint8 * max(1, len(clear)))(*clear) if clear is not None else None
        args = (self.engine.handle, name.encode(), _ptr(rows), a_cap, count, _ptr(enc[0]) if enc else None, enc[1] if enc else 0,
                buf, len(clear) if clear is not None else 0)
        n_out = C.c_uint32(0)
        _check(lib().fhe_str_op_many(*args, None, C.byref(n_out)))          # outputs per row (builds and caches the plan)
        out = self._alloc((count, n_out.value, big))
        _check(lib().fhe_str_op_many(*args, _ptr(out), C.byref(n_out)))
        return out

    def eq_many(self, rows, b): return self.op_many("eq", rows, b)[:, 0]
    def ne_many(self, rows, b): return self.op_many("ne", rows, b)[:, 0]
    def contains_many(self, rows, b): return self.op_many("contains", rows, b)[:, 0]
    def starts_with_many(self, rows, b): return self.op_many("starts_with", rows, b)[:, 0]
    def ends_with_many(self, rows, b): return self.op_many("ends_with", rows, b)[:, 0]
    def find_many(self, rows, b): return self.op_many("find", rows, b)

    def eq(self, a, b): return self._binary("eq", a, b)[0]
    def ne(self, a, b): return self._binary("ne", a, b)[0]
    def starts_with(self, a, b): return self._binary("starts_with", a, b)[0]
    def ends_with(self, a, b): return self._binary("ends_with", a, b)[0]
    def contains(self, a, b): return self._binary("contains", a, b)[0]
    def find(self, a, b): return self._binary("find", a, b)
    def rfind(self, a, b): return self._binary("rfind", a, b)
    def eq_ignore_case(self, a, b): return self._binary("eq_ignore_case", a, b)[0]
    def lt(self, a, b): return self._binary("lt", a, b)[0]
    def le(self, a, b): return self._binary("le", a, b)[0]
    def gt(self, a, b): return self._binary("gt", a, b)[0]
    def ge(self, a, b): return self._binary("ge", a, b)[0]

    def _n_digits(self, cap):
        n = 0
        while (self.engine.params.msg_mod ** n) < cap + 1:
            n += 1
        return n

    def len(self, a):
        a, a_cap = self._cap(a)
        out = self._alloc((self._n_digits(a_cap), self.engine.params.big_size))
        _check(lib().fhe_str_len(self.engine.handle, _ptr(a), a_cap, _ptr(out)))
        return out

    def is_empty(self, a):
        a, a_cap = self._cap(a)
        out = self._alloc((1, self.engine.params.big_size))
        _check(lib().fhe_str_is_empty(self.engine.handle, _ptr(a), a_cap, _ptr(out)))
        return out[0]

    def _strip_affix(self, op, a, pat):
        """pat: clear bytes, or an encrypted (zero padded) pattern."""
        a, a_cap = self._cap(a)
        out = self._alloc((1 + a.shape[0], self.engine.params.big_size))
        if isinstance(pat, (bytes, bytearray)):
            buf = (C.c_uint8 * max(1, len(pat)))(*pat)
            _check(getattr(lib(), f"fhe_str_{op}_clear")(self.engine.handle, _ptr(a), a_cap, buf, len(pat), _ptr(out)))
        else:
            pat, p_cap = self._cap(pat)
            _check(getattr(lib(), f"fhe_str_{op}")(self.engine.handle, _ptr(a), a_cap, _ptr(pat), p_cap, _ptr(out)))
        return out[0], out[1:]

    def strip_prefix(self, a, pat): return self._strip_affix("strip_prefix", a, pat)
    def strip_suffix(self, a, pat): return self._strip_affix("strip_suffix", a, pat)

    def _unary(self, op, a):
        a, a_cap = self._cap(a)
        out = self._alloc(a.shape)
        _check(getattr(lib(), f"fhe_str_{op}")(self.engine.handle, _ptr(a), a_cap, _ptr(out)))
        return out

    def trim_start(self, a): return self._unary("trim_start", a)
    def trim_end(self, a): return self._unary("trim_end", a)
    def strip(self, a): return self._unary("strip", a)

    def replace(self, a, frm, to, out_cap: int | None = None):
        """Replace every leftmost non-overlapping occurrence of frm by to (bytes.replace).  frm / to: both
        clear bytes, or both encrypted strings.  out_cap=None: the equal-length in-place form (encrypted
        operands unpadded).  With out_cap: any lengths, encrypted operands may be zero padded, the result
        has out_cap characters."""
        a, a_cap = self._cap(a)
        big = self.engine.params.big_size
        clear = isinstance(frm, (bytes, bytearray))
        if out_cap is None:
            out = self._alloc(a.shape)
            if clear:
                if len(frm) != len(to):
                    raise FheError("replace: `from` and `to` of different lengths need an output capacity (out_cap)")
                fb = (C.c_uint8 * max(1, len(frm)))(*frm)
                tb = (C.c_uint8 * max(1, len(to)))(*to)
                _check(lib().fhe_str_replace_clear(self.engine.handle, _ptr(a), a_cap, fb, tb, len(frm), _ptr(out)))
            else:
                frm, f_cap = self._cap(frm)
                to, t_cap = self._cap(to)
                if f_cap != t_cap:
                    raise FheError("replace: `from` and `to` of different capacities need an output capacity (out_cap)")
                both = np.concatenate([frm, to])
                _check(lib().fhe_str_replace(self.engine.handle, _ptr(a), a_cap, _ptr(both), f_cap, _ptr(out)))
            return out
        out = self._alloc((out_cap * self.bpc, big))
        if clear:
            fb = (C.c_uint8 * max(1, len(frm)))(*frm)
            tb = (C.c_uint8 * max(1, len(to)))(*to)
            _check(lib().fhe_str_replace_clear_general(self.engine.handle, _ptr(a), a_cap, fb, len(frm), tb, len(to),
                                                       out_cap, _ptr(out)))
        else:
            frm, f_cap = self._cap(frm)
            to, t_cap = self._cap(to)
            _check(lib().fhe_str_replace_general(self.engine.handle, _ptr(a), a_cap, _ptr(frm), f_cap,
                                                 _ptr(to) if t_cap else None, t_cap, out_cap, _ptr(out)))
        return out

    def concat(self, a, b):
        """a ++ b (padding of a removed); b encrypted (any capacity) or clear bytes."""
        a, a_cap = self._cap(a)
        if isinstance(b, (bytes, bytearray)):
            out = self._alloc(((a_cap + len(b)) * self.bpc, self.engine.params.big_size))
            buf = (C.c_uint8 * max(1, len(b)))(*b)
            _check(lib().fhe_str_concat_clear(self.engine.handle, _ptr(a), a_cap, buf, len(b), _ptr(out)))
        else:
            b, b_cap = self._cap(b)
            out = self._alloc(((a_cap + b_cap) * self.bpc, self.engine.params.big_size))
            _check(lib().fhe_str_concat(self.engine.handle, _ptr(a), a_cap, _ptr(b), b_cap, _ptr(out)))
        return out

    def repeat(self, a, count: int):
        a, a_cap = self._cap(a)
        out = self._alloc((count * a_cap * self.bpc, self.engine.params.big_size))
        _check(lib().fhe_str_repeat_clear(self.engine.handle, _ptr(a), a_cap, count, _ptr(out)))
        return out

    def to_upper(self, a): return self._unary("to_upper", a)
    def to_lower(self, a): return self._unary("to_lower", a)
