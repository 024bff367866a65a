"""ctypes loader for libdsabf.so.  Fails loudly: there is no CPU or PyTorch fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# DSABF_LIB_PATH: a measurement build made by tools/build_variant.py (A/B experiments only; never set by tests or bench.py)
LIB_PATH = os.environ.get("DSABF_LIB_PATH") or os.path.join(PKG, "libdsabf.so")

BF_OK = 0
BF_NOT_READY = 1
BF_ERR_INVALID, BF_ERR_DEVICE, BF_ERR_NO_DEVICE, BF_ERR_STATE = -1, -2, -3, -4
BF_DETECT_CANONICAL, BF_DETECT_FAST, BF_DETECT_CONTRACTED = 0, 1, 2


class BfConfig(C.Structure):
    """Mirror of ``bf_config`` (include/dsabf.h)."""

    _fields_ = [(n, C.c_int) for n in ("n_beams", "n_ant", "n_freq", "n_pol", "n_avg", "n_out_per_gemm",
                                       "n_gemms_per_block", "n_blocks_on_gpu", "n_streams", "verbose", "detect_mode")]


class BfhEventOps(C.Structure):
    """Mirror of ``bfh_event_ops`` (include/dsabf_host.h): an event backend as a table of C callbacks."""

    CREATE = C.CFUNCTYPE(C.c_void_p, C.c_void_p)
    DESTROY = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    RECORD = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
    QUERY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
    _fields_ = [("user", C.c_void_p), ("create", CREATE), ("destroy", DESTROY), ("record_transfer", RECORD),
                ("record_analysis", RECORD), ("query", QUERY)]


class BfGatherMsg(C.Structure):
    """Mirror of ``bf_gather_msg`` (include/dsabf.h)."""

    _fields_ = [("kind", C.c_int), ("peer", C.c_int), ("local_offset", C.c_size_t), ("full_offset", C.c_size_t),
                ("count", C.c_size_t)]


class DsabfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libdsabf error %d: %s" % (code, msg))
        self.code = code


_lib = None

# name -> (restype, argtypes); every symbol include/dsabf.h declares
SIGNATURES = {
    "bf_last_error": (C.c_char_p, []),
    "bf_version": (C.c_char_p, []),
    "bf_config_default": (C.c_int, [C.POINTER(BfConfig), C.c_int]),
    "bf_n_inputs_per_output": (C.c_int, [C.POINTER(BfConfig)]),
    "bf_n_timesteps_per_gemm": (C.c_int, [C.POINTER(BfConfig)]),
    "bf_bytes_per_gemm": (C.c_size_t, [C.POINTER(BfConfig)]),
    "bf_bytes_per_block": (C.c_size_t, [C.POINTER(BfConfig)]),
    "bf_floats_per_detect": (C.c_size_t, [C.POINTER(BfConfig)]),
    "bf_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bf_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "bf_create": (C.c_int, [C.POINTER(BfConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "bf_destroy": (C.c_int, [C.c_void_p]),
    "bf_get_config": (C.c_int, [C.c_void_p, C.POINTER(BfConfig)]),
    "bf_set_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bf_set_weights_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bf_alloc_pinned": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "bf_free_pinned": (C.c_int, [C.c_void_p]),
    "bf_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "bf_host_unregister": (C.c_int, [C.c_void_p]),
    "bf_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "bf_event_create_on": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "bf_event_destroy": (C.c_int, [C.c_void_p]),
    "bf_event_query": (C.c_int, [C.c_void_p]),
    "bf_event_synchronize": (C.c_int, [C.c_void_p]),
    "bf_submit_block": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bf_record_transfer_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bf_enqueue_gemm_unit": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bf_enqueue_block": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bf_enqueue_block_dedisperse": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bf_enqueue_dedisperse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bf_record_analysis_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bf_stream_sync": (C.c_int, [C.c_void_p, C.c_int]),
    "bf_timer_start": (C.c_int, [C.c_void_p]),
    "bf_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "bf_beamform_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bf_expand_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "bf_gemm_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bf_dedisperse_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bf_dedisperse_dm_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p]),
    "bf_dm_stream_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bf_dm_stream_destroy": (C.c_int, [C.c_void_p]),
    "bf_dm_stream_max_delay": (C.c_int, [C.c_void_p]),
    "bf_dm_stream_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.c_void_p]),
    "bf_dm_stream_output_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "bf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "bf_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "bf_comm_destroy": (C.c_int, [C.c_void_p]),
    "bf_comm_rank": (C.c_int, [C.c_void_p]),
    "bf_comm_world": (C.c_int, [C.c_void_p]),
    "bf_comm_library_info": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "bf_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "bf_gather_detected": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bf_gather_detected_staged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bf_gather_offset": (C.c_size_t, [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_size_t]),
    "bf_gather_rows_held": (C.c_size_t, [C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "bf_gather_plan": (C.c_size_t, [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(BfGatherMsg),
                                    C.c_size_t]),
    "bf_block_output_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "bf_queue_stream": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "bf_block_gather_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bf_block_gather_stage_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bf_enqueue_d2h": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bf_dedisperse_band_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bf_dedisperse_dm_band_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p]),
    "bf_gather_relayout_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_void_p]),
    "bf_set_switch": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "bf_get_counter": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]),
    "bf_rtw_plan": (C.c_int, [C.POINTER(BfConfig), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bf_kernel_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)]),
    "bf_kernel_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "bf_mfma_peak_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double),
                                    C.c_void_p]),
    "bf_launch_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.c_char_p, C.c_size_t]),
    "bf_enqueue_block_to": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bf_dm_stream_reserve": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "bf_variant_key": (C.c_int, [C.POINTER(BfConfig), C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "bf_handle_variant_key": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]),
    # ---- include/dsabf_host.h ----
    "bfh_default_positions": (C.c_int, [C.c_int, C.c_void_p]),
    "bfh_default_directions": (C.c_int, [C.c_int, C.c_void_p]),
    "bfh_read_positions": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p]),
    "bfh_read_directions": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p]),
    "bfh_count_entries": (C.c_int, [C.c_char_p]),
    "bfh_write_python_file": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "bfh_channel_frequency": (C.c_float, [C.c_int, C.c_int, C.c_int]),
    "bfh_make_weights": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bfh_gen_create": (C.c_int, [C.POINTER(BfConfig), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bfh_gen_destroy": (C.c_int, [C.c_void_p]),
    "bfh_gen_read_sources": (C.c_int, [C.c_void_p, C.c_char_p]),
    "bfh_gen_set_sources": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "bfh_gen_generate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "bfh_gen_data": (C.c_void_p, [C.c_void_p]),
    "bfh_gen_size": (C.c_size_t, [C.c_void_p]),
    "bfh_gen_n_pt_sources": (C.c_int, [C.c_void_p]),
    "bfh_gen_need_more": (C.c_int, [C.c_void_p, C.c_int]),
    "bfh_gen_ready": (C.c_int, [C.c_void_p, C.c_int]),
    "bfh_obs_create": (C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(BfConfig), C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "bfh_obs_destroy": (C.c_int, [C.c_void_p]),
    "bfh_obs_generate_transfer_event": (C.c_int, [C.c_void_p]),
    "bfh_obs_generate_analysis_event": (C.c_int, [C.c_void_p]),
    "bfh_obs_check_transfer_events": (C.c_int, [C.c_void_p]),
    "bfh_obs_check_analysis_events": (C.c_int, [C.c_void_p]),
    "bfh_obs_counters": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4),
    "bfh_obs_check_ready_for_transfer": (C.c_int, [C.c_void_p]),
    "bfh_obs_check_ready_for_analysis": (C.c_int, [C.c_void_p]),
    "bfh_obs_check_ready_for_dh2_transfer": (C.c_int, [C.c_void_p, C.c_int]),
    "bfh_obs_check_observations_complete": (C.c_int, [C.c_void_p]),
    "bfh_obs_check_transfers_complete": (C.c_int, [C.c_void_p]),
    "bfh_obs_set_transfers_complete": (C.c_int, [C.c_void_p, C.c_int]),
    "bfh_obs_set_n_pt_sources": (C.c_int, [C.c_void_p, C.c_int]),
    "bfh_obs_get_current_analysis_gemm": (C.c_uint64, [C.c_void_p, C.c_int]),
    "bfh_obs_get_current_transfer_gemm": (C.c_uint64, [C.c_void_p]),
    "bfh_obs_get_next_gpu_analysis_block": (C.c_uint64, [C.c_void_p]),
    "bfh_obs_get_next_gpu_transfer_block": (C.c_uint64, [C.c_void_p]),
    "bfh_obs_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "bfh_obs_create_custom": (C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(BfConfig), C.POINTER(BfhEventOps), C.c_int,
                                        C.POINTER(C.c_void_p)]),
    "bfh_obs_status": (C.c_int, [C.c_void_p]),
    "bfh_run_observation_junk": (C.c_int, [C.POINTER(BfConfig), C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]),
    "bfh_run_observation_junk_to_file": (C.c_int, [C.POINTER(BfConfig), C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_float),
                                                   C.POINTER(C.c_uint64), C.c_void_p]),
    "bfh_run_observation_junk_dm": (C.c_int, [C.POINTER(BfConfig), C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_float),
                                              C.POINTER(C.c_uint64), C.c_void_p]),
    "bfh_run_observation_junk_sharded": (C.c_int, [C.POINTER(BfConfig), C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p,
                                                   C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.c_void_p]),
    "bfh_run_observation_junk_to_ring": (C.c_int, [C.POINTER(BfConfig), C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                                   C.c_char_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint64),
                                                   C.c_void_p]),
    "bfh_file_sink_create": (C.c_int, [C.POINTER(BfConfig), C.c_char_p, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "bfh_sink_acquire": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(C.c_float))]),
    "bfh_sink_commit": (C.c_int, [C.c_void_p, C.c_uint64]),
    "bfh_sink_close": (C.c_int, [C.c_void_p]),
    "bfh_sink_destroy": (C.c_int, [C.c_void_p]),
    "bfh_dm_trials": (C.c_int, [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.POINTER(C.c_double), C.c_int]),
    "bfh_dm_delays": (C.c_int, [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_double, C.c_double,
                                C.POINTER(C.c_int32)]),
    "bfh_dm_sink_create": (C.c_int, [C.POINTER(BfConfig), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bfh_dm_sink_deliver": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bfh_dm_sink_destroy": (C.c_int, [C.c_void_p]),
    "bfh_dm_trial_share": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bfh_junk_fill": (C.c_int, [C.POINTER(BfConfig), C.c_int, C.c_uint64, C.c_void_p]),
    "bfh_shm_ring_create": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.POINTER(C.c_void_p)]),
    "bfh_shm_ring_attach": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "bfh_shm_ring_detach": (C.c_int, [C.c_void_p]),
    "bfh_shm_ring_unlink": (C.c_int, [C.c_char_p]),
    "bfh_shm_ring_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "bfh_shm_ring_write": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "bfh_shm_ring_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bfh_run_observation_shm_dm": (C.c_int, [C.POINTER(BfConfig), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p,
                                             C.c_int, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_int)]),
    "bfh_run_observation_shm": (C.c_int, [C.POINTER(BfConfig), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                          C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "bfh_run_debug_observation": (C.c_int, [C.POINTER(BfConfig), C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                            C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                            C.POINTER(C.c_float)]),
    "bfh_run_debug_observation2": (C.c_int, [C.POINTER(BfConfig), C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                             C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                             C.POINTER(C.c_float), C.c_int]),
}


_hip = None
_rccl_path = None


def _preload_hip_runtime():
    """libdsabf.so carries no DT_NEEDED for the HIP runtime (see build.py): make exactly ONE libamdhip64 visible
    process-wide before loading it -- torch's bundled copy when torch is installed (so that torch streams, events
    and allocations belong to the same runtime as our launches), else the system ROCm one."""
    global _hip
    if _hip is not None:
        return _hip
    cands = []
    try:
        import torch  # noqa: F401  (plumbing only: device memory, streams, torch.distributed)

        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except ImportError:
        pass
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so"), "libamdhip64.so"]
    errs = []
    for c in cands:
        if os.path.sep in c and not os.path.exists(c):
            continue
        try:
            _hip = C.CDLL(c, mode=C.RTLD_GLOBAL)
            # RCCL must come from the same place as the HIP runtime (bf_comm.cpp binds it with dlopen at first use):
            # torch's bundled librccl.so next to torch's libamdhip64.so, ROCm's next to ROCm's.
            global _rccl_path
            for r in ("librccl.so", "librccl.so.1"):
                rp = os.path.join(os.path.dirname(c), r) if os.path.sep in c else r
                if os.path.sep in rp and os.path.exists(rp):
                    # only a file that exists: a bare soname would turn bf_comm.cpp's soft resolution order (a copy
                    # already in the process, librccl.so.1, librccl.so) into its hard-fail DSABF_RCCL_LIB branch
                    _rccl_path = rp
                    os.environ.setdefault("DSABF_RCCL_LIB", rp)
                    break
            return _hip
        except OSError as e:  # pragma: no cover
            errs.append("%s: %s" % (c, e))
    raise ImportError("no HIP runtime (libamdhip64.so) could be loaded: %s" % "; ".join(errs))


def load() -> C.CDLL:
    """Load libdsabf.so (built by ``dsabeamformer_amd.build.build()`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -m dsabeamformer_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback for the beamformer hot path." % LIB_PATH)
        _preload_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
            except AttributeError:
                if os.environ.get("DSABF_LIB_PATH"):   # a measurement build of another round beside the product (tools/ab_libs.py):
                    continue                           # it may predate an export; calling that one fails, the rest works
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise DsabfError(rc, load().bf_last_error().decode(errors="replace"))
    return rc
