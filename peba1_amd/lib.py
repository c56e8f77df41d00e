"""ctypes loader for libtfhe-hip.so (the C-ABI drop-in boundary, include/*.h).

The library is the product: it fails loudly if the shared object is missing and
the library itself aborts if no HIP device is present when a gate is evaluated.
There is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PEBA1_TFHE_HIP_LIB: another build of the same library (tools/diag/build_variants.sh: A/B runs of kernel variants on one
# box); it must exist like the default one -- there is no fallback either way
LIB_PATH = os.environ.get("PEBA1_TFHE_HIP_LIB") or os.path.join(_HERE, "libtfhe-hip.so")
CIRCUITS_PATH = os.path.join(_HERE, "libpeba1-circuits.so")


class LweSample(C.Structure):
    _fields_ = [("a", C.POINTER(C.c_int32)), ("b", C.c_int32), ("slot", C.c_int32),
                ("current_variance", C.c_double)]


class LweParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("alpha_min", C.c_double), ("alpha_max", C.c_double)]


class TLweParams(C.Structure):
    _fields_ = [("N", C.c_int32), ("k", C.c_int32), ("alpha_min", C.c_double), ("alpha_max", C.c_double)]


class TGswParams(C.Structure):
    _fields_ = [("l", C.c_int32), ("Bgbit", C.c_int32), ("Bg", C.c_int32), ("halfBg", C.c_int32),
                ("maskMod", C.c_uint32), ("tlwe_params", C.POINTER(TLweParams)), ("kpl", C.c_int32),
                ("offset", C.c_uint32)]


class ParameterSet(C.Structure):
    _fields_ = [("ks_t", C.c_int32), ("ks_basebit", C.c_int32), ("in_out_params", C.POINTER(LweParams)),
                ("tgsw_params", C.POINTER(TGswParams))]


class CloudKeySet(C.Structure):
    _fields_ = [("params", C.POINTER(ParameterSet)), ("bk", C.c_void_p), ("bkFFT", C.c_void_p)]


class SecretKeySet(C.Structure):
    _fields_ = [("params", C.POINTER(ParameterSet)), ("lwe_key", C.c_void_p), ("tgsw_key", C.c_void_p),
                ("cloud", CloudKeySet)]


class Stats(C.Structure):
    _fields_ = [("blind_rotates", C.c_uint64), ("keyswitches", C.c_uint64), ("linear_ops", C.c_uint64),
                ("levels", C.c_uint64), ("flushes", C.c_uint64), ("br_launches", C.c_uint64),
                ("ms_blind_rotate", C.c_double), ("ms_keyswitch", C.c_double), ("ms_flush_wall", C.c_double),
                ("ms_blind_rotate_busy", C.c_double), ("reused_gates", C.c_uint64),
                ("br8_launches", C.c_uint64), ("br8_rotations", C.c_uint64), ("ms_blind_rotate8", C.c_double),
                ("clk_shader_cycles", C.c_uint64), ("clk_ref_ticks", C.c_uint64), ("dead_gates", C.c_uint64),
                ("folded_gates", C.c_uint64)]


PS = C.POINTER(ParameterSet)
CK = C.POINTER(CloudKeySet)
SK = C.POINTER(SecretKeySet)
LS = C.POINTER(LweSample)
I32P = C.POINTER(C.c_int32)

# every symbol include/tfhe/tfhe_gate_bootstrapping_functions.h and include/tfhe_hip.h declare
_GATE2 = ["bootsAND", "bootsOR", "bootsXOR", "bootsXNOR", "bootsNAND", "bootsNOR", "bootsANDNY", "bootsANDYN",
          "bootsORNY", "bootsORYN"]
SIGNATURES = {
    "new_gate_bootstrapping_ciphertext_array": (LS, [C.c_int32, PS]),
    "delete_gate_bootstrapping_ciphertext_array": (None, [C.c_int32, LS]),
    "new_gate_bootstrapping_ciphertext": (LS, [PS]),
    "delete_gate_bootstrapping_ciphertext": (None, [LS]),
    "bootsCONSTANT": (None, [LS, C.c_int32, CK]),
    "bootsNOT": (None, [LS, LS, CK]),
    "bootsCOPY": (None, [LS, LS, CK]),
    "bootsMUX": (None, [LS, LS, LS, LS, CK]),
    "new_default_gate_bootstrapping_parameters": (PS, [C.c_int32]),
    "new_random_gate_bootstrapping_secret_keyset": (SK, [PS]),
    "delete_gate_bootstrapping_parameters": (None, [PS]),
    "delete_gate_bootstrapping_secret_keyset": (None, [SK]),
    "delete_gate_bootstrapping_cloud_keyset": (None, [CK]),
    "bootsSymEncrypt": (None, [LS, C.c_int32, SK]),
    "bootsSymDecrypt": (C.c_int32, [LS, SK]),
    "modSwitchFromTorus32": (C.c_int32, [C.c_int32, C.c_int32]),
    "modSwitchToTorus32": (C.c_int32, [C.c_int32, C.c_int32]),
    "export_tfheGateBootstrappingParameterSet_toFile": (None, [C.c_void_p, PS]),
    "new_tfheGateBootstrappingParameterSet_fromFile": (PS, [C.c_void_p]),
    "export_tfheGateBootstrappingCloudKeySet_toFile": (None, [C.c_void_p, CK]),
    "new_tfheGateBootstrappingCloudKeySet_fromFile": (CK, [C.c_void_p]),
    "export_tfheGateBootstrappingSecretKeySet_toFile": (None, [C.c_void_p, SK]),
    "new_tfheGateBootstrappingSecretKeySet_fromFile": (SK, [C.c_void_p]),
    "export_gate_bootstrapping_ciphertext_toFile": (None, [C.c_void_p, LS, PS]),
    "import_gate_bootstrapping_ciphertext_fromFile": (None, [C.c_void_p, LS, PS]),
    "tfhe_hip_test_wg_times": (C.c_int, [CK, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "tfhe_hip_last_error": (C.c_char_p, []),
    "tfhe_hip_clear_error": (None, []),
    "tfhe_hip_set_device": (C.c_int, [C.c_int]),
    "tfhe_hip_get_device": (C.c_int, []),
    "tfhe_hip_device_pci_bus_id": (C.c_int, [C.c_char_p, C.c_int]),
    "tfhe_hip_new_parameters": (PS, [C.c_int32] * 7 + [C.c_double] * 3),
    "tfhe_hip_new_p2048_parameters": (PS, []),
    "tfhe_hip_new_secret_keyset_seeded": (SK, [PS, C.c_uint64]),
    "tfhe_hip_new_secret_keyset_seeded_host": (SK, [PS, C.c_uint64]),
    "tfhe_hip_set_encrypt_seed": (None, [C.c_uint64]),
    "tfhe_hip_randomness_is_seeded": (C.c_int, []),
    "tfhe_hip_test_chacha20_block": (None, [C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "tfhe_hip_key_lwe": (I32P, [SK, C.POINTER(C.c_int64)]),
    "tfhe_hip_key_tlwe": (I32P, [SK, C.POINTER(C.c_int64)]),
    "tfhe_hip_key_bk": (I32P, [CK, C.POINTER(C.c_int64)]),
    "tfhe_hip_key_ksk": (I32P, [CK, C.POINTER(C.c_int64)]),
    "tfhe_hip_sample_words": (C.c_int32, [PS]),
    "tfhe_hip_export_samples": (C.c_int, [LS, C.c_int32, PS, I32P]),
    "tfhe_hip_import_samples": (C.c_int, [LS, C.c_int32, PS, I32P]),
    "tfhe_hip_export_samples_device": (C.c_int, [LS, C.c_int32, PS, C.c_void_p]),
    "tfhe_hip_import_samples_device": (C.c_int, [LS, C.c_int32, PS, C.c_void_p]),
    "tfhe_hip_export_samples_device_async": (C.c_int, [LS, C.c_int32, PS, C.c_void_p]),
    "tfhe_hip_import_samples_device_async": (C.c_int, [LS, C.c_int32, PS, C.c_void_p]),
    "tfhe_hip_stream": (C.c_void_p, []),
    "tfhe_hip_sync_samples": (C.c_int, [LS, C.c_int32]),
    "tfhe_hip_set_deferred": (None, [C.c_int]),
    "tfhe_hip_get_deferred": (C.c_int, []),
    "tfhe_hip_flush": (C.c_int, []),
    "tfhe_hip_flush_async": (C.c_int, []),
    "tfhe_hip_wait": (C.c_int, []),
    "tfhe_hip_stream_sync": (C.c_int, []),
    "tfhe_hip_wait_event": (C.c_int, [C.c_void_p, C.c_char_p]),
    "tfhe_hip_set_diag_label": (None, [C.c_char_p]),
    "tfhe_hip_gate_batch": (C.c_int, [C.c_int, LS, LS, LS, C.c_int32, CK]),
    "tfhe_hip_set_tuning": (C.c_int, [C.c_char_p, C.c_int64]),
    "tfhe_hip_test_form_admissible": (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int]),
    "tfhe_hip_test_set_alloc_cap": (None, [C.c_int64]),
    "tfhe_hip_get_stats": (None, [C.POINTER(Stats)]),
    "tfhe_hip_reset_stats": (None, []),
    "tfhe_hip_set_kernel_timing": (None, [C.c_int]),
    "tfhe_hip_test_schedule": (C.c_int, [I32P, C.c_int32, C.c_int32, C.c_int32, I32P]),
    "tfhe_hip_kernel_negacyclic": (C.c_int, [CK, I32P, I32P, I32P, C.c_int32]),
    "tfhe_hip_kernel_bootstrap_woks": (C.c_int, [CK, I32P, C.c_int32, I32P, I32P]),
    "tfhe_hip_kernel_keyswitch": (C.c_int, [CK, I32P, C.c_int32, I32P]),
}
for _g in _GATE2:
    SIGNATURES[_g] = (None, [LS, LS, LS, CK])
_lib = None


def load():
    """Load libtfhe-hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(peba1_amd has no CPU fallback)")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)   # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib

