"""ctypes binding of include/enspara_hip.h (libenspara_hip.so, in-tree).

There is no CPU fallback: if the library is missing, or no HIP device is
visible when a device call is made, this raises.  torch is imported first so
that the library binds to the HIP runtime torch already loaded (both carry
the soname libamdhip64.so.7); torch itself is only plumbing here
(torch.distributed for the multi-GPU exchange).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ENSPARA_HIP_LIB",
                          os.path.join(HERE, "libenspara_hip.so"))

EK_TILE = 256
EK_OK, EK_EARG, EK_EHIP, EK_ESTATE, EK_ENOMEM = 0, -1, -2, -3, -4

# every symbol include/enspara_hip.h declares (tests check the export table)
SYMBOLS = [
    "ek_abi_version", "ek_last_error", "ek_device_count",
    "ek_ctx_create", "ek_ctx_destroy", "ek_ctx_sync", "ek_ctx_stream",
    "ek_load_frames", "ek_rmsd_to_center",
    "ek_state_reset", "ek_state_download", "ek_state_upload",
    "ek_record_bytes", "ek_local_candidate", "ek_own_record",
    "ek_kcenters_step", "ek_kcenters_run",
    "ek_history_download", "ek_history_reset",
    "ek_spec_candidates", "ek_round_candidates", "ek_quad_copy_ready",
    "ek_spec_begin", "ek_spec_round", "ek_spec_localmax",
    "ek_spec_apply", "ek_spec_round_end", "ek_spec_progress", "ek_spec_rounds",
    "ek_spec_chain_bytes", "ek_spec_chain_rows", "ek_spec_chain_max",
    "ek_spec_chain_apply", "ek_run_stats", "ek_ti_stats",
    "ek_ms_setup", "ek_ms_mailbox", "ek_ms_connect", "ek_ms_begin", "ek_ms_local",
    "ek_ms_global", "ek_ms_end", "ek_ms_run", "ek_ms_state", "ek_ms_diag",
    "ek_assign_nearest",
    "ek_pam_begin", "ek_pam_count_members", "ek_pam_select_member",
    "ek_pam_propose", "ek_pam_propose_member", "ek_pam_commit",
    "ek_pam_count_members_batch", "ek_pam_select_members_batch",
    "ek_pam_prefetch", "ek_pam_propose_ex", "ek_pam_prefetch_stats",
    "ek_pam_prefetch_window", "ek_pam_prefetch_passes", "ek_pam_sparse_stats", "ek_pam_ahead_stats", "ek_pam_window_run", "ek_pam_sweep", "ek_np_choice_draws",
    "ek_pam_window_max",
    "ek_pam_prefetch_centers_window",
    "ek_centered_frames", "ek_pam_begin_table", "ek_pam_prefetch_centers",
    "ek_pam_propose_center",
    "ek_msm_counts", "ek_msm_counts_ctx", "ek_msm_row_normalize",
    "ek_krylov_create", "ek_krylov_destroy", "ek_krylov_set_vector",
    "ek_krylov_get_vector", "ek_krylov_step", "ek_krylov_rotate",
    "ek_krylov_combine", "ek_krylov_expand", "ek_krylov_set_filter",
    "ek_feat_create", "ek_feat_destroy", "ek_feat_load", "ek_feat_distance",
    "ek_feat_kcenters", "ek_feat_pam_sweep", "ek_feat_pam_release",
    "ek_set_frames_per_lane", "ek_set_option", "ek_get_option", "ek_last_run_timing",
    "ek_reserve_centers",
    "ek_debug_guards",
    "ek_timing_begin", "ek_timing_end", "ek_timing_form", "ek_hbm_copy_rate",
    "ek_qcp_probe",
]


# enum ek_option of include/enspara_hip.h (tests/test_abi.py holds the two equal)
EK_OPT_NONTEMPORAL = 1
EK_OPT_ASSIGN_KERNEL = 2
EK_OPT_CANDIDATES = 4
EK_OPT_CHAINED = 5
EK_OPT_PAM_PRUNE = 6
EK_OPT_STATE_EXACT = 7
EK_OPT_ADAPTIVE = 8
EK_OPT_PASS_FORM = 9
EK_OPT_FUSED_ROUNDS = 10
EK_OPT_TRIANGLE = 11
EK_OPT_PAM_ONE_WORKGROUP = 12
EK_OPT_PAM_MAX_PAIRS = 13
EK_OPT_PAM_BOTH_SUMS = 14
EK_OPT_FINE_PICK = 15
EK_OPT_PAM_BOUNDS = 16
EK_OPT_PICK_CAP = 17
EK_OPT_SMALL_SHARDS = 18
EK_OPT_PAM_AHEAD = 19
EK_OPT_PAM_ZERO_COPY = 20
EK_OPT_PAM_PAIRS_MFMA = 21
EK_OPT_PASS_SWEEP = 22
EK_OPT_MS_TWO_PHASE = 23
OPTIONS = {k[7:].lower(): v for k, v in list(globals().items())
           if k.startswith("EK_OPT_")}


class HipLibraryMissing(RuntimeError):
    pass


class HipError(RuntimeError):
    pass


_lib = None


def load():
    """Load libenspara_hip.so (once).  Raises HipLibraryMissing if it has not
    been built (python -m enspara_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found: the HIP extension is not built "
            "(run `python -m enspara_amd.build`); there is no CPU fallback."
            % LIB_PATH)
    if os.environ.get("ENSPARA_NO_TORCH") != "1":   # (measurement switch: the system's
        try:                                        # HIP runtime instead of torch's copy)
            import torch  # noqa: F401  (loads libamdhip64.so.7 first)
        except Exception:  # pragma: no cover - torch is plumbing, not required
            pass
    try:
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise HipLibraryMissing("cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    f32p, i32p, i64p = (C.POINTER(C.c_float), C.POINTER(C.c_int32),
                        C.POINTER(C.c_int64))
    L.ek_abi_version.restype = C.c_int
    L.ek_last_error.restype = C.c_char_p
    L.ek_device_count.restype = C.c_int
    L.ek_ctx_create.argtypes = [C.c_int, i64, i32, i64, vp, C.POINTER(vp)]
    L.ek_ctx_destroy.argtypes = [vp]
    L.ek_ctx_sync.argtypes = [vp]
    L.ek_ctx_stream.restype = vp
    L.ek_ctx_stream.argtypes = [vp]
    L.ek_load_frames.argtypes = [vp, vp, i64, i64, C.c_int]
    L.ek_rmsd_to_center.argtypes = [vp, i64, f32p, f32p]
    L.ek_state_reset.argtypes = [vp]
    L.ek_state_download.argtypes = [vp, f32p, i32p]
    L.ek_state_upload.argtypes = [vp, f32p, i32p]
    L.ek_record_bytes.restype = C.c_size_t
    L.ek_record_bytes.argtypes = [i32]
    L.ek_local_candidate.argtypes = [vp, vp]
    L.ek_own_record.restype = vp
    L.ek_own_record.argtypes = [vp]
    L.ek_kcenters_step.argtypes = [vp, vp, i32, i32, C.c_double, vp]
    L.ek_kcenters_run.argtypes = [vp, i32, i32, C.c_double, i32p, i64p, f32p, f32p]
    L.ek_history_download.argtypes = [vp, i32, i32, i64p, f32p, i32p]
    L.ek_history_reset.argtypes = [vp]
    L.ek_spec_candidates.argtypes = [vp]
    L.ek_round_candidates.argtypes = [vp]
    L.ek_quad_copy_ready.argtypes = [vp]
    L.ek_spec_begin.argtypes = [vp, i32, i32, vp]
    L.ek_spec_round.argtypes = [vp, vp, i32, C.c_double]
    L.ek_spec_localmax.argtypes = [vp, vp]
    L.ek_spec_apply.argtypes = [vp, vp, i32, C.c_double]
    L.ek_spec_chain_bytes.argtypes = [i32p, i32p]
    L.ek_spec_chain_rows.argtypes = [vp, vp]
    L.ek_spec_chain_max.argtypes = [vp, vp, i32, vp]
    L.ek_spec_chain_apply.argtypes = [vp, vp, i32, C.c_double]
    L.ek_spec_round_end.argtypes = [vp, vp]
    L.ek_spec_progress.argtypes = [vp, i32p, i32p]
    L.ek_spec_rounds.argtypes = [vp, i32p]
    L.ek_run_stats.argtypes = [vp, i64p, i64p]
    L.ek_ms_setup.argtypes = [vp, i32, i32, C.POINTER(C.c_size_t)]
    L.ek_ms_mailbox.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), vp, vp]
    L.ek_ms_connect.argtypes = [vp, i32, vp, vp, vp, vp]
    L.ek_ms_begin.argtypes = [vp, i32, i32]
    L.ek_ms_local.argtypes = [vp, C.c_double, vp]
    L.ek_ms_global.argtypes = [vp, C.c_double, vp]
    L.ek_ms_end.argtypes = [vp]
    L.ek_ms_state.argtypes = [vp, i32p, i32p, i32p]
    L.ek_ms_run.argtypes = [vp, i32, i32, C.c_double, i32p, i64p, f32p, f32p]
    L.ek_ti_stats.argtypes = [vp, i64p, i64p]
    L.ek_assign_nearest.argtypes = [vp, f32p, i32]
    f64p = C.POINTER(C.c_double)
    L.ek_ms_diag.argtypes = [vp, i64p, f64p]
    L.ek_pam_begin.argtypes = [vp, i64p, i32]
    L.ek_pam_count_members.argtypes = [vp, i32, i64p]
    L.ek_pam_select_member.argtypes = [vp, i32, i64, i64p]
    L.ek_pam_propose.argtypes = [vp, i32, i64, f64p, f64p, i64p]
    L.ek_pam_propose_member.argtypes = [vp, i32, i64, i64p, f64p, f64p, i64p]
    L.ek_pam_commit.argtypes = [vp, C.c_int]
    L.ek_pam_count_members_batch.argtypes = [vp, i32, i32, i64p]
    L.ek_pam_select_members_batch.argtypes = [vp, i32, i32, i64p, i64p]
    L.ek_pam_prefetch.argtypes = [vp, i64p, i32]
    L.ek_pam_propose_ex.argtypes = [vp, i32, i64, i64, i32, i32, f64p, f64p,
                                    i64p, C.POINTER(C.c_uint32)]
    L.ek_pam_prefetch_stats.argtypes = [vp, i64p, i64p]
    L.ek_pam_window_run.argtypes = [vp, i32, i32, i64p, i64p, i32, i32, i32p, i32p,
                                    f64p, f64p, i64p]
    L.ek_pam_window_max.argtypes = []
    L.ek_pam_window_max.restype = i32
    L.ek_pam_prefetch_window.argtypes = [vp, i64p, i32, i32, i32]
    L.ek_pam_prefetch_passes.argtypes = [vp, i64p, i64p]
    L.ek_pam_sparse_stats.argtypes = [vp, i64p, i64p]
    L.ek_pam_ahead_stats.argtypes = [vp, i64p]
    L.ek_np_choice_draws.argtypes = [C.POINTER(C.c_uint32), i64, i64p, i64p, i64, i64p]
    L.ek_np_choice_draws.restype = i64
    L.ek_pam_sweep.argtypes = [vp, i32, i32, C.POINTER(C.c_uint32), i64, i64p, i64p, i32p, i64p,
                               i32p, f64p, f64p, i64p, i32p]
    L.ek_pam_prefetch_centers_window.argtypes = [vp, vp, vp, i32, i32, i32]
    L.ek_centered_frames.argtypes = [vp, i64p, i32p, i32, vp, vp]
    L.ek_pam_begin_table.argtypes = [vp, vp, vp, i32]
    L.ek_pam_prefetch_centers.argtypes = [vp, vp, vp, i32]
    L.ek_pam_propose_center.argtypes = [vp, i32, i32, vp, vp, i64, i32, i32, vp]
    L.ek_msm_counts.argtypes = [C.c_int, i32p, i64p, i64, i32, i32, i32, i64,
                                i32p, i32p, i64p, i64p]
    L.ek_msm_counts_ctx.argtypes = [vp, i64p, i64, i32, i32, i32, i64, i32p, i32p,
                                    i64p, i64p]
    L.ek_msm_row_normalize.argtypes = [C.c_int, i64p, f64p, i64, f64p, f64p]
    L.ek_krylov_create.argtypes = [C.c_int, i64, i64p, i32p, f64p, i32,
                                   C.POINTER(vp)]
    L.ek_krylov_destroy.argtypes = [vp]
    L.ek_krylov_set_vector.argtypes = [vp, i32, f64p]
    L.ek_krylov_get_vector.argtypes = [vp, i32, f64p]
    L.ek_krylov_step.argtypes = [vp, i32, i32, f64p]
    L.ek_krylov_set_filter.argtypes = [vp, i32, C.c_double, C.c_double]
    L.ek_krylov_rotate.argtypes = [vp, i32, i32, f64p, i32]
    L.ek_krylov_combine.argtypes = [vp, i32, i32, f64p, f64p]
    L.ek_krylov_expand.argtypes = [vp, i32, i32, f64p, i32]
    L.ek_feat_create.argtypes = [C.c_int, i64, i32, i32, C.POINTER(vp)]
    L.ek_feat_destroy.argtypes = [vp]
    L.ek_feat_load.argtypes = [vp, vp, i64, i64]
    L.ek_feat_distance.argtypes = [vp, i32, vp, f64p]
    L.ek_feat_pam_sweep.argtypes = [vp, i32, i32, i64p, i64p, C.POINTER(C.c_uint32), i64,
                                    i64p, f64p, i32p, i32p, i32p, i32p]
    L.ek_feat_pam_release.argtypes = [vp]
    L.ek_feat_pam_release.restype = None
    L.ek_feat_kcenters.argtypes = [vp, i32, i32, i32, C.c_double, f64p, i32p, i64p,
                                   i32p, f64p]
    L.ek_set_frames_per_lane.argtypes = [vp, C.c_int]
    L.ek_set_option.argtypes = [vp, i32, i32]
    L.ek_get_option.argtypes = [vp, i32, i32p]
    L.ek_reserve_centers.argtypes = [vp, i32]
    if os.environ.get("ENSPARA_HIP_LIB") is None or hasattr(L, "ek_debug_guards"):
        L.ek_debug_guards.argtypes = [vp]     # (a variant library from before it: tools/)
    L.ek_last_run_timing.argtypes = [vp, f32p, i32p]
    L.ek_timing_begin.argtypes = [vp, i32, i32]
    L.ek_timing_end.argtypes = [vp, f32p, i32p]
    L.ek_timing_form.argtypes = [vp, i32p]
    L.ek_hbm_copy_rate.argtypes = [C.c_int, C.c_size_t, f64p]
    L.ek_qcp_probe.argtypes = [C.c_int, vp, vp, vp, i32, vp, C.c_int64, vp, vp, vp]
    for name in SYMBOLS:
        if name == "ek_debug_guards" and os.environ.get("ENSPARA_HIP_LIB"):
            continue        # (a variant library from before it: tools/)
        getattr(L, name)
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().ek_last_error().decode("utf-8", "replace")
        raise HipError("enspara_hip error %d: %s" % (rc, msg))


def f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))
