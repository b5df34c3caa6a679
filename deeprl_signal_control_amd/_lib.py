"""ctypes loader for the C-ABI library (include/tsc.h).  There is no CPU fallback: if
libtsc.so is missing or fails to load, importing the product path raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# TSC_LIB: a measurement build of the same sources (tools/build_variant.sh: libtsc_<name>.so next to the library), A/B runs only
LIB_PATH = os.environ.get('TSC_LIB') or os.path.join(HERE, 'libtsc.so')

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class TscScenario(C.Structure):
    _fields_ = [
        ('n_lane', C.c_int32), ('n_route', C.c_int32), ('n_agent', C.c_int32), ('n_flow', C.c_int32),
        ('k_max', C.c_int32), ('p_max', C.c_int32), ('l_max', C.c_int32), ('s_max', C.c_int32),
        ('nbr_max', C.c_int32),
        ('lane_len', _fp), ('lane_vmax', _fp), ('lane_det_start', _fp),
        ('lane_node', _ip), ('lane_up', _ip),
        ('mv_next', _ip), ('mv_link', _ip), ('mv_yield', _ip), ('mv_prio', _ip), ('mv_zip', _ip), ('route_entry', _ip), ('flows', _ip),
        ('agent_lanes', _ip), ('agent_nlane', _ip), ('agent_nlink', _ip), ('agent_nphase', _ip),
        ('green_tab', _bp), ('yellow_tab', _bp),
        ('nbr', _ip), ('obs_kind', _ip), ('obs_src', _ip),
        ('control_interval_sec', C.c_int32), ('yellow_interval_sec', C.c_int32),
        ('episode_length_sec', C.c_int32), ('teleport_sec', C.c_int32),
        ('queue_cap', C.c_int32), ('objective', C.c_int32), ('agent_kind', C.c_int32),
        ('realnet_scale', C.c_int32),
        ('coop_gamma', C.c_double), ('norm_wave', C.c_double), ('norm_wait', C.c_double),
        ('clip_wave', C.c_double), ('clip_wait', C.c_double), ('coef_wait', C.c_double),
        ('lane_origin', _fp),
        ('n_stream', C.c_int32), ('k_choice', C.c_int32),
        ('stream_entry', _ip), ('stream_origin', _fp), ('stream_limit', _fp), ('stream_mode', _ip), ('stream_choice', _ip),
        ('n_interval', C.c_int32), ('choice_interval_sec', C.c_int32),
        ('lane_sib', _ip),
    ]


_LIB = None

# every symbol include/tsc.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = ['tsc_last_error', 'tsc_version', 'tsc_profile_enable', 'tsc_profile_select', 'tsc_profile_reset', 'tsc_profile_read',
           'tsc_profile_name', 'tsc_env_create', 'tsc_env_destroy', 'tsc_env_set_stream', 'tsc_env_set_resident_instances',
           'tsc_env_reset', 'tsc_env_set_stream_routes', 'tsc_env_set_greedy', 'tsc_env_greedy_actions', 'tsc_env_set_fingerprint', 'tsc_env_bind_fingerprint', 'tsc_env_reward_sum', 'tsc_env_step', 'tsc_env_get_state',
           'tsc_env_live_vehicles', 'tsc_env_vehicle_counts', 'tsc_env_set_block_order', 'tsc_env_counters', 'tsc_env_debug_clock', 'tsc_env_live_sum', 'tsc_env_record', 'tsc_env_read_record', 'tsc_env_read_trips',
           'tsc_model_create', 'tsc_model_destroy', 'tsc_model_set_stream', 'tsc_model_layout',
           'tsc_model_set_params', 'tsc_model_reset_opt_state', 'tsc_model_debug_read', 'tsc_model_get_params', 'tsc_model_get_opt_state', 'tsc_model_set_opt_state',
           'tsc_model_reset', 'tsc_model_forward', 'tsc_model_forward_sample', 'tsc_model_sample', 'tsc_model_add_transition',
           'tsc_model_rollout_slot', 'tsc_model_compute_grads', 'tsc_model_grad_buffer', 'tsc_model_apply_grads', 'tsc_model_get_returns', 'tsc_model_debug_clock',
           'tsc_gemm_grouped_f32',
           'tsc_iql_create', 'tsc_iql_destroy', 'tsc_iql_set_stream', 'tsc_iql_layout', 'tsc_iql_set_params', 'tsc_iql_get_params',
           'tsc_iql_get_opt_state', 'tsc_iql_set_opt_state', 'tsc_iql_forward', 'tsc_iql_add_transition', 'tsc_iql_replay_size',
           'tsc_iql_compute_grads', 'tsc_iql_compute_grads_at', 'tsc_iql_grad_buffer', 'tsc_iql_apply_grads', 'tsc_iql_debug_batch', 'tsc_iql_path', 'tsc_iql_debug_clock']


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('deeprl_signal_control_amd: %s not built -- run `python -c "import __graft_entry__ as g; '
                           'g.build()"` (hipcc, gfx950). There is no CPU fallback.' % LIB_PATH)
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so, same soname as /opt/rocm's); it must be
    # the one in the process, so load it first and let libtsc.so bind to it.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.tsc_last_error.restype = C.c_char_p
    L.tsc_profile_name.restype = C.c_char_p
    L.tsc_profile_name.argtypes = [C.c_int32]
    L.tsc_profile_enable.argtypes = [C.c_int32]
    L.tsc_profile_select.argtypes = [C.c_uint64]
    L.tsc_profile_read.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.tsc_env_create.argtypes = [C.POINTER(TscScenario), C.c_int32, C.c_int32, C.POINTER(vp)]
    L.tsc_env_destroy.argtypes = [vp]
    L.tsc_env_set_stream.argtypes = [vp, vp]
    L.tsc_env_set_resident_instances.argtypes = [vp, C.c_int32]
    L.tsc_env_reset.argtypes = [vp, C.POINTER(C.c_uint32), vp]
    L.tsc_env_set_stream_routes.argtypes = [vp, _ip]
    L.tsc_env_set_fingerprint.argtypes = [vp, vp]
    L.tsc_env_set_greedy.argtypes = [vp, C.c_int32, C.c_int32, _ip, _ip, _ip]
    L.tsc_env_greedy_actions.argtypes = [vp, vp, vp]
    L.tsc_env_bind_fingerprint.argtypes = [vp, vp]
    L.tsc_env_reward_sum.argtypes = [vp, C.POINTER(C.c_double), C.c_int32]
    L.tsc_env_step.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32]
    L.tsc_env_get_state.argtypes = [vp, C.c_int32, _ip, _fp, _fp, _fp, _ip, _ip, _ip, _ip, _ip]
    L.tsc_env_live_vehicles.argtypes = [vp, C.POINTER(C.c_double)]
    L.tsc_env_vehicle_counts.argtypes = [vp, vp]
    L.tsc_env_set_block_order.argtypes = [vp, vp]
    L.tsc_env_counters.argtypes = [vp, vp, vp]
    L.tsc_env_debug_clock.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64)]
    L.tsc_env_live_sum.argtypes = [vp, C.POINTER(C.c_double), C.c_int32]
    L.tsc_env_record.argtypes = [vp, C.c_int32, C.c_int32]
    L.tsc_env_read_record.argtypes = [vp, vp, vp, vp]
    L.tsc_env_read_trips.argtypes = [vp, C.c_int32, vp, C.c_int32, C.POINTER(C.c_int32)]
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError('libtsc: ' + lib().tsc_last_error().decode())


def scenario_struct(scn):
    """Pack a scenario.Scenario into the C struct; returns (struct, keepalive list)."""
    keep = []

    def arr(a, dt, ptr):
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data_as(ptr)

    A = scn.n_agent
    nbr_max = max(1, max(len(n) for n in scn.neighbors))
    nbr = np.full((A, nbr_max), -1, np.int32)
    for a, ns in enumerate(scn.neighbors):
        nbr[a, :len(ns)] = ns
    agent_kind = {'greedy': 0, 'a2c': 0, 'ma2c': 2}.get(scn.agent, 1)
    s = TscScenario(
        n_lane=scn.n_lane, n_route=scn.n_route, n_agent=A, n_flow=len(scn.flows),
        k_max=scn.green_tab.shape[2], p_max=scn.green_tab.shape[1], l_max=scn.agent_lanes.shape[1],
        s_max=scn.s_max, nbr_max=nbr_max,
        lane_len=arr(scn.lane_len, np.float32, _fp), lane_vmax=arr(scn.lane_vmax, np.float32, _fp),
        lane_det_start=arr(scn.lane_det_start, np.float32, _fp),
        lane_node=arr(scn.lane_node, np.int32, _ip),
        lane_up=arr(scn.lane_up, np.int32, _ip), mv_next=arr(scn.mv_next, np.int32, _ip),
        mv_link=arr(scn.mv_link, np.int32, _ip), mv_yield=arr(scn.mv_yield, np.int32, _ip),
        mv_prio=arr(scn.mv_prio, np.int32, _ip), mv_zip=arr(scn.mv_zip, np.int32, _ip), route_entry=arr(scn.route_entry_lane, np.int32, _ip),
        flows=arr(scn.flows, np.int32, _ip), agent_lanes=arr(scn.agent_lanes, np.int32, _ip),
        agent_nlane=arr(scn.agent_nlane, np.int32, _ip), agent_nlink=arr(scn.agent_nlink, np.int32, _ip),
        agent_nphase=arr(scn.agent_nphase, np.int32, _ip),
        green_tab=arr(scn.green_tab, np.uint8, _bp), yellow_tab=arr(scn.yellow_tab, np.uint8, _bp),
        nbr=arr(nbr, np.int32, _ip), obs_kind=arr(scn.obs_kind, np.int32, _ip),
        obs_src=arr(scn.obs_src, np.int32, _ip),
        control_interval_sec=scn.control_interval_sec, yellow_interval_sec=scn.yellow_interval_sec,
        episode_length_sec=scn.episode_length_sec, teleport_sec=scn.teleport_sec,
        queue_cap=scn.queue_cap, objective={'queue': 0, 'wait': 1, 'hybrid': 2}[scn.objective],
        agent_kind=agent_kind, realnet_scale=int(scn.reward_scale_realnet),
        coop_gamma=scn.coop_gamma, norm_wave=scn.norm_wave, norm_wait=scn.norm_wait,
        clip_wave=scn.clip_wave, clip_wait=scn.clip_wait, coef_wait=scn.coef_wait,
        lane_origin=arr(scn.lane_origin, np.float32, _fp))
    if getattr(scn, 'lane_sib', None) is not None:
        s.lane_sib = arr(scn.lane_sib, np.int32, _ip)
    scn.streams_ready()
    scn.check_limits()
    if scn.stream_entry_lane is not None:
        s.n_stream, s.k_choice = int(scn.n_stream), int(scn.stream_choice.shape[2])
        s.n_interval, s.choice_interval_sec = int(scn.stream_choice.shape[1]), int(min(scn.choice_interval_sec, 1 << 30))
        s.stream_limit = arr(scn.stream_limit, np.float32, _fp)
        s.stream_entry = arr(scn.stream_entry_lane, np.int32, _ip)
        s.stream_origin = arr(scn.stream_origin, np.float32, _fp)
        s.stream_mode = arr(scn.stream_mode, np.int32, _ip)
        s.stream_choice = arr(scn.stream_choice, np.int32, _ip)
    return s, keep


def profile_names():
    L = lib()
    out, i = [], 0
    while True:
        name = L.tsc_profile_name(i).decode()
        if not name:
            return out
        out.append(name)
        i += 1


def profile_select(names=None):
    """Bracket only these kernels with event pairs (None = all): an event pair between two dependent launches inflates the
    FOLLOWING launch's figure, so a kernel is measured cleanly with only its own launches bracketed."""
    mask = 0
    if names:
        all_names = profile_names()
        for n in names:
            mask |= 1 << all_names.index(n)
    check(lib().tsc_profile_select(mask))


def profile(enable=None, reset=False):
    """Per-kernel HIP-event timings of the library: returns {name: (total_ms, count)}.
    enable: False/0 off, True/1 every launch, n > 1 every n-th launch of the per-control-step kernels
    (total_ms is then average x launches, see include/tsc.h)."""
    L = lib()
    if enable is not None:
        check(L.tsc_profile_enable(int(enable)))
    if reset:
        check(L.tsc_profile_reset())
        return {}
    out, i = {}, 0
    while True:
        name = L.tsc_profile_name(i).decode()
        if not name:
            break
        ms, cnt = C.c_double(), C.c_int64()
        check(L.tsc_profile_read(i, C.byref(ms), C.byref(cnt)))
        if cnt.value:
            out[name] = (ms.value, cnt.value)
        i += 1
    return out
