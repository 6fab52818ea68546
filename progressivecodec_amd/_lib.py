"""ctypes binding of libpcodec.so (include/pcodec.h).  There is no fallback: if the
library is missing the import fails, and creating a codec without a HIP device fails."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PC_LIB") or os.path.join(_HERE, "libpcodec.so")   # PC_LIB: another build of the same library (same-box A/B)

PC_OK = 0
ERRORS = {-1: "PC_ERR_ARG", -2: "PC_ERR_INDEX", -3: "PC_ERR_BUFFER", -4: "PC_ERR_TRUNCATED", -5: "PC_ERR_CDF",
          -6: "PC_ERR_HIP", -7: "PC_ERR_NOMEM", -8: "PC_ERR_STATE", -9: "PC_ERR_MISSING"}

#: every symbol include/pcodec.h declares (tests check the library exports all of them)
EXPORTS = [
    "pc_version", "pc_strerror", "pc_last_hip_error", "pc_rans_bound", "pc_rans_encode_with_indexes",
    "pc_rans_decode_with_indexes", "pc_rans_encode_batch", "pc_rans_decode_batch", "pc_pmf_to_quantized_cdf",
    "pc_pack_conv_weight", "pc_conv2d_nhwc", "pc_gdn_nhwc", "pc_win_attention_nhwc", "pc_mask_quantile_threshold",
    "pc_gc_prep_encode", "pc_gc_prep_decode_index", "pc_gc_dequantize", "pc_codec_create", "pc_codec_destroy",
    "pc_codec_set_tensor", "pc_codec_set_tables", "pc_codec_finalize", "pc_codec_set_threads", "pc_codec_compress",
    "pc_codec_num_slices", "pc_codec_get_string", "pc_codec_decompress", "pc_codec_read_tap", "pc_codec_read_tap_i32",
    "pc_codec_profile_begin", "pc_codec_profile_end", "pc_codec_compress_levels", "pc_codec_get_level_string",
    "pc_codec_decompress_levels", "pc_codec_forward", "pc_codec_set_cust_map", "pc_codec_strings_size", "pc_codec_copy_strings",
    "pc_codec_decompress_packed", "pc_host_pool_plan", "pc_rans_decode_stream", "pc_codec_set_scale_table", "pc_codec_profile_bytes", "pc_contract_id", "pc_rans_decode_batch_u8", "pc_codec_set_rem",
    "pc_codec_host_stats", "pc_codec_set_rem_checkpoint", "pc_codec_set_option", "pc_selftest_packed_gelu", "pc_profile_set_epoch", "pc_codec_profile_intervals",
]


class PcodecError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = lib().pc_strerror(code).decode() if _lib is not None else ""
        hip = lib().pc_last_hip_error() if (_lib is not None and code == -6) else 0
        super().__init__(f"{where}: {ERRORS.get(code, code)} ({msg})" + (f" hipError={hip}" if hip else ""))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  progressivecodec_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        vp, i32p, f32p, u8p, sz = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_size_t
        L.pc_version.restype = C.c_char_p
        L.pc_contract_id.restype = C.c_uint32
        L.pc_strerror.restype = C.c_char_p
        L.pc_strerror.argtypes = [C.c_int]
        L.pc_rans_bound.restype = sz
        L.pc_rans_bound.argtypes = [sz]
        L.pc_rans_encode_with_indexes.argtypes = [vp, vp, sz, vp, C.c_int, C.c_int, vp, vp, vp, sz, C.POINTER(sz)]
        L.pc_rans_decode_with_indexes.argtypes = [vp, sz, vp, sz, vp, C.c_int, C.c_int, vp, vp, vp]
        L.pc_rans_decode_batch_u8.argtypes = [vp, vp, sz, vp, sz, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int]
        L.pc_rans_decode_stream.argtypes = [vp, sz, vp, vp, sz, vp, C.c_int, C.c_int, vp, vp, vp]
        L.pc_host_pool_plan.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pc_rans_encode_batch.argtypes = [vp, vp, sz, sz, vp, C.c_int, C.c_int, vp, vp, vp, sz, vp, C.c_int]
        L.pc_rans_decode_batch.argtypes = [vp, vp, sz, vp, sz, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int]
        L.pc_pmf_to_quantized_cdf.argtypes = [vp, C.c_int, C.c_int, vp]
        L.pc_pack_conv_weight.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        L.pc_conv2d_nhwc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, vp, vp]
        L.pc_gdn_nhwc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]
        L.pc_win_attention_nhwc.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.pc_mask_quantile_threshold.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
        L.pc_gc_prep_encode.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int,
                                        vp, C.c_int, C.c_float, vp, vp, vp, vp, C.c_int, vp]
        L.pc_gc_prep_decode_index.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_float, vp, vp, vp]
        L.pc_gc_dequantize.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]
        L.pc_codec_create.argtypes = [C.POINTER(vp), C.c_int]
        L.pc_codec_destroy.argtypes = [vp]
        L.pc_codec_destroy.restype = None
        L.pc_codec_set_tensor.argtypes = [vp, C.c_char_p, vp, C.c_int, C.POINTER(C.c_int64), C.c_int]
        L.pc_codec_set_tables.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
        L.pc_codec_finalize.argtypes = [vp]
        L.pc_codec_set_threads.argtypes = [vp, C.c_int]
        L.pc_codec_set_option.argtypes = [vp, C.c_char_p, C.c_int]
        if hasattr(L, "pc_selftest_packed_gelu"):           # (absent from older builds selected with PC_LIB for a same-box A/B)
            L.pc_selftest_packed_gelu.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pc_codec_compress.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp]
        L.pc_codec_num_slices.argtypes = [vp]
        L.pc_codec_get_string.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(sz)]
        L.pc_codec_decompress.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp]
        L.pc_codec_compress_levels.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]
        L.pc_codec_get_level_string.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(sz)]
        L.pc_codec_decompress_levels.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]
        L.pc_codec_set_cust_map.argtypes = [vp, vp]
        L.pc_codec_set_rem.argtypes = [vp, vp, C.c_int]
        L.pc_codec_set_rem_checkpoint.argtypes = [vp, vp]
        L.pc_codec_strings_size.argtypes = [vp, C.POINTER(sz), C.POINTER(C.c_int)]
        L.pc_codec_copy_strings.argtypes = [vp, vp, sz, vp, sz]
        L.pc_codec_decompress_packed.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]
        L.pc_codec_forward.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp, C.c_int, vp]
        L.pc_codec_set_scale_table.argtypes = [vp, vp, C.c_int]
        L.pc_codec_read_tap.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(sz)]
        L.pc_codec_read_tap_i32.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(sz)]
        L.pc_codec_profile_begin.argtypes = [vp]
        L.pc_profile_set_epoch.argtypes = [C.c_int]
        L.pc_codec_profile_intervals.argtypes = [vp, vp, vp, vp, sz, C.POINTER(sz)]
        L.pc_codec_profile_bytes.argtypes = [vp, C.POINTER(C.c_double)]
        L.pc_codec_host_stats.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
        L.pc_codec_profile_end.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def check(code, where=""):
    if code != PC_OK:
        raise PcodecError(code, where)
