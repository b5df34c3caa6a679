"""Scenario compiler: turns a traffic scenario into the dense tables the HIP
microsimulator, the CPU oracle and the host-side env mirror all consume.

Only ``large_grid`` is generated from first principles here; every rule is a
restatement of the reference's own network generator and env class:

* node grid / boundary nodes ........ large_grid/data/build_file.py:27-50
* road types (2-lane 20 m/s streets, 1-lane 11 m/s avenues) ... :53-58
* edges (200 m internal, 75 m boundary) ........................ :14-19,66-98
* lane-to-lane connections (dedicated left lane on streets) .... :107-124
* lane-area detectors ``pos=-50 endPos=-1`` on incoming lanes ... :360-391,445
* demand (12 OD pairs, piece-wise constant vehsPerHour) ........ :268-326
* phase set, neighbour map ........ envs/large_grid_env.py:38-42,73-101
* agent order = sorted node ids, lanes = dedup of controlled lanes in link
  order ........................................ envs/env.py:207-254
* state/fingerprint dims ........................ envs/env.py:303-323
* yellow-phase rule .............................. envs/env.py:128-152

The SUMO-internal parts the reference does not contain (route choice, lane
choice, junction right-of-way) are *defined* here and in MICROSIM_SPEC.md ("microsim
spec"); they are this repo's spec, not SUMO's.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from fractions import Fraction
from typing import Dict, List, Tuple

import numpy as np

# --- microsim spec constants (MICROSIM_SPEC.md) -------------------
VEH_LEN = 5.0        # vType length=5            (build_file.py:279)
VEH_ACCEL = 5.0      # vType accel=5
VEH_DECEL = 10.0     # vType decel=10
MIN_GAP = 2.5        # SUMO default minGap (not in the reference tree): sizes lane pieces and contracted chains
STAND_GAP = 2.0      # standstill gap of the microsim spec (csrc/tsc_env.hip kS0; MICROSIM_SPEC.md)
LANE_CAP = 28        # vehicle slots per lane: 200 m / 7.5 m = 26.7 -> 27 (+1); hand-offs stop at LANE_CAP - MAX_CROSS
LANE_CHANGE_DEFAULT = True    # large_grid: lane choice by the junction's connections + lane changes on the two-lane streets (rule 10)
MAX_CROSS = 4        # vehicles that may leave one lane in one sim-step
MAX_UP = 4           # upstream feeder lanes per lane
DET_LEN = 50.0       # lane-area detector covers the last 50 m
NO_LANE = -1


@dataclass
class Scenario:
    name: str
    agent: str                       # 'ma2c' | 'ia2c' | 'greedy' | ...
    node_names: List[str]            # sorted TL ids == agent order
    n_agent: int
    # lanes
    lane_names: List[str]
    lane_len: np.ndarray             # f32 [NL]
    lane_vmax: np.ndarray            # f32 [NL]
    lane_node: np.ndarray            # i32 [NL] agent index of downstream TL node or -1 (sink)
    lane_det_start: np.ndarray       # f32 [NL] detector start position (L-50, or 0 = whole lane)
    lane_up: np.ndarray              # i32 [NL, MAX_UP] feeder lanes (ascending) or -1
    # routes
    n_route: int
    mv_next: np.ndarray              # i32 [NL, NR] next lane, -1 = arrive at lane end, -2 = n/a
    mv_link: np.ndarray              # i32 [NL, NR] link index at downstream node (3*approach+mv) or -1
    mv_yield: np.ndarray             # i32 [NL, NR] lane whose head this movement yields to, or -1
    mv_prio: np.ndarray              # i32 [NL, NR] 1 = priority movement (others may have to yield to it)
    mv_zip: np.ndarray               # i32 [NL, NR] zipper merge slot: rank | count << 8 (0 = no zipper)
    route_entry_lane: np.ndarray     # i32 [NR]
    route_names: List[Tuple[str, str]]
    # signals
    agent_lanes: np.ndarray          # i32 [A, LMAX] incoming lanes in ild order, -1 pad
    agent_nlane: np.ndarray          # i32 [A]
    agent_nlink: np.ndarray          # i32 [A]
    agent_nphase: np.ndarray         # i32 [A]   (= n_a)
    link_lane: np.ndarray            # i32 [A, KMAX] incoming lane of every signal link
    phases: List[List[str]]          # per agent list of phase strings
    green_tab: np.ndarray            # u8 [A, PMAX, KMAX]  chars of phase p
    yellow_tab: np.ndarray           # u8 [A, PMAX, PMAX, KMAX] chars for prev->new during yellow
    # neighbourhood
    neighbors: List[List[int]]       # agent indices, neighbour-list order
    n_s_ls: List[int]
    n_w_ls: List[int]
    n_f_ls: List[int]
    n_a_ls: List[int]
    obs_kind: np.ndarray             # i32 [A, SMAX] 0 pad,1 wave,2 wave*coop,3 wait,4 fingerprint
    obs_src: np.ndarray              # i32 [A, SMAX] lane id (1,2,3) or agent*AMAX+k (4)
    # demand
    flows: np.ndarray                # i32 [NF, 4] begin,end,vph,route
    obs_len: List[int] = field(default_factory=list)   # true obs length per agent (greedy: wave only)
    # env constants (ENV_CONFIG)
    control_interval_sec: int = 5
    yellow_interval_sec: int = 2
    episode_length_sec: int = 3600
    coop_gamma: float = 0.9
    norm_wave: float = 5.0
    norm_wait: float = 100.0
    clip_wave: float = 2.0
    clip_wait: float = 2.0
    coef_wait: float = 0.2
    objective: str = 'hybrid'
    has_wait_state: bool = True
    queue_cap: int = -1              # real_net: min(10, halting)   (env.py:332-333)
    reward_scale_realnet: bool = False
    teleport_sec: int = 600          # --time-to-teleport (env.py:281-284)
    extra: Dict = field(default_factory=dict)
    link_foes: np.ndarray = None     # u32 [A, KMAX] bit k2 of row (a, k): the path of signal link k2 crosses or joins the path of link k
                                     # inside the junction (junction interiors, MICROSIM_SPEC.md rule 10); None = no junction has foes
    lane_sib: np.ndarray = None      # i32 [NL] the other lane of a two-lane street, -1 = none (None: the scenario has no lane changing,
                                     # MICROSIM_SPEC.md rule 10): a vehicle on a lane that does not serve its movement moves over to it
    lane_origin: np.ndarray = None   # f32 [NL] where the SUMO lane of that name begins inside the compiled lane (0 unless
                                     # contract_chains merged upstream pieces into it): `lane.*` TraCI getters count from here

    # insertion streams (include/tsc.h tsc_scenario): None = every route is its own stream (flows[:, 3] = route)
    stream_entry_lane: np.ndarray = None    # i32 [NS]
    stream_origin: np.ndarray = None        # f32 [NS] start of the insertion window on the entry lane
    stream_limit: np.ndarray = None         # f32 [NS] its end (the SUMO entry lane may be one piece of a contracted chain)
    stream_mode: np.ndarray = None          # i32 [NS] 0 fixed route, 1 per-vehicle draw (turn ratios), 2 per-episode route from the host
    stream_choice: np.ndarray = None        # i32 [NS, NI, KC, 2] (route, cumulative weight of 65536) per choice interval, route -1 pads
    choice_interval_sec: int = 1 << 30      # length of a choice interval (time-variant turn ratios); NI = stream_choice.shape[1]

    def __post_init__(self):
        if self.lane_origin is None:
            self.lane_origin = np.zeros(len(self.lane_names), np.float32)
        if self.link_foes is None:
            self.link_foes = np.zeros(np.asarray(self.link_lane).shape, np.uint32)

    @property
    def n_stream(self) -> int:
        return self.n_route if self.stream_entry_lane is None else len(self.stream_entry_lane)

    def greedy_controller_tables(self):
        """What the reference's greedy controller of this scenario holds, as tables for tsc_env_set_greedy: per agent the
        candidates it compares, each a list of indices into the agent's observation (its own wave entries come first in every
        layout) summed IN THIS ORDER, and the action a winning candidate stands for.  Returns (n_cand i32 [A], term i32
        [A, GC, GT] -1-padded, action i32 [A, GC]).
        large_grid: LargeGridController.greedy's hard-coded lane pairs (envs/large_grid_env.py:56-60), candidate = phase.
        small_grid: SmallGridController.greedy compares the first len(STATE_PHASE_MAP[node]) entries and returns the mapped
        phase (envs/small_grid_env.py:29-30,51-55).
        otherwise (real_net): RealNetController.greedy (envs/real_net_env.py:90-111) -- per phase the lanes of its 'G' links
        ('g' does not count) in link order, every lane once."""
        A = self.n_agent
        cands, acts = [], []
        if self.name == 'large_grid':
            pairs = [(0, 3), (2, 5), (1, 4), (1, 2), (4, 5)]
            cands, acts = [[list(p) for p in pairs]] * A, [list(range(5))] * A
        elif self.name == 'small_grid':
            spm = self.extra['state_phase_map']
            for n in self.node_names:
                cands.append([[k] for k in range(len(spm[n]))])
                acts.append([int(x) for x in spm[n]])
        else:
            for a in range(A):
                lanes = [int(x) for x in self.agent_lanes[a, :self.agent_nlane[a]]]
                ca = []
                for p in range(int(self.agent_nphase[a])):
                    seen = []
                    for k in range(int(self.agent_nlink[a])):
                        if self.green_tab[a, p, k] == ord('G'):
                            j = lanes.index(int(self.link_lane[a, k]))
                            if j not in seen:
                                seen.append(j)
                    ca.append(seen)
                cands.append(ca)
                acts.append(list(range(len(ca))))
        GC = max(len(c) for c in cands)
        GT = max(1, max(len(t) for c in cands for t in c))
        n_cand = np.array([len(c) for c in cands], np.int32)
        term = np.full((A, GC, GT), -1, np.int32)
        action = np.zeros((A, GC), np.int32)
        for a in range(A):
            for c, t in enumerate(cands[a]):
                term[a, c, :len(t)] = t
                action[a, c] = acts[a][c]
        return n_cand, term, action

    def streams_ready(self):
        """Fill the optional stream tables (whole-lane insertion window) once streams are declared."""
        if self.stream_entry_lane is not None:
            NS = len(self.stream_entry_lane)
            if self.stream_origin is None:
                self.stream_origin = np.zeros(NS, np.float32)
            if self.stream_limit is None:
                self.stream_limit = np.asarray(self.lane_len, np.float32)[np.asarray(self.stream_entry_lane)]
            if self.stream_mode is None:
                self.stream_mode = np.zeros(NS, np.int32)
            assert self.stream_choice.ndim == 4 and self.stream_choice.shape[0] == NS and self.stream_choice.shape[3] == 2
        return self

    def check_limits(self):
        """The fixed capacities of csrc/tsc_env.hip (include/tsc.h TSC_LANE_CAP / TSC_MAX_UP / TSC_MAX_CROSS): a scenario that
        does not fit is refused here with the remedy, not truncated on the device."""
        veh = VEH_LEN + MIN_GAP          # sizing rule; hand-offs into a lane stop at LANE_CAP - MAX_CROSS vehicles whatever its length
        long_ = [self.lane_names[l] for l in range(self.n_lane) if int(self.lane_len[l] // veh) > LANE_CAP]
        if long_:
            raise ValueError('lanes %s hold more than %d standing vehicles (%.1f m each): split them into pieces of at most %.0f m '
                             '(as build_small_grid does with its 400 m edges)' % (long_[:4], LANE_CAP, veh, LANE_CAP * veh))
        if self.n_route > 254 or self.n_stream > 254:
            raise ValueError('%d routes / %d insertion streams: at most 254 each (route ids and stream lists are bytes on the device)'
                             % (self.n_route, self.n_stream))
        if self.n_lane > 255 and (np.asarray(self.mv_zip) >> 8 > 1).any():
            raise ValueError('%d lanes with zipper merges: the merge arbitration keeps a lane\'s feeders as bytes on the device, '
                             'lane indices must stay below 255' % self.n_lane)
        if np.asarray(self.lane_up).shape[1] != MAX_UP:
            raise ValueError('lane_up must list %d feeder slots per lane' % MAX_UP)
        if self.control_interval_sec > 8:
            raise ValueError('control_interval_sec %d > 8 unsupported' % self.control_interval_sec)
        return self

    def entry_lanes(self):
        """(entry lane, route) of every way a vehicle can enter: what decides which lanes can ever be occupied."""
        if self.stream_entry_lane is None:
            return [(int(l), r) for r, l in enumerate(self.route_entry_lane)]
        return [(int(self.stream_entry_lane[s]), int(r)) for s in range(self.n_stream)
                for r in sorted({int(x) for x in self.stream_choice[s, :, :, 0].ravel() if x >= 0})]

    @property
    def n_lane(self) -> int:
        return len(self.lane_names)

    @property
    def s_max(self) -> int:
        return int(self.obs_kind.shape[1])

    @property
    def a_max(self) -> int:
        return int(max(self.n_a_ls))


# ---------------------------------------------------------------------------
# reference rules restated
# ---------------------------------------------------------------------------
def yellow_phase(prev_phase: str, cur_phase: str) -> str:
    """Link states shown during the yellow interval when switching
    prev_phase -> cur_phase (envs/env.py:137-152)."""
    switch_reds, switch_greens = [], []
    for i, (p0, p1) in enumerate(zip(prev_phase, cur_phase)):
        if (p0 in 'Gg') and (p1 == 'r'):
            switch_reds.append(i)
        elif (p0 in 'r') and (p1 in 'Gg'):
            switch_greens.append(i)
    if not switch_reds:
        return cur_phase
    y = list(cur_phase)
    for i in switch_reds:
        y[i] = 'y'
    for i in switch_greens:
        y[i] = 'r'
    return ''.join(y)


LARGE_GRID_PHASES = ['GGgrrrGGgrrr', 'rrrGrGrrrGrG', 'rrrGGrrrrGGr',
                     'rrrGGGrrrrrr', 'rrrrrrrrrGGG']   # envs/large_grid_env.py:40-41


def large_grid_demand(peak_flow1: int, peak_flow2: int):
    """(from_edge, to_edge, begin, end, vph) for the 84 flow elements, in file
    order (large_grid/data/build_file.py:284-324).  vph keeps the reference's
    ``%d`` truncation (e.g. 1100*0.6*0.7 -> 461)."""
    edge_maps = [0, 1, 2, 3, 4, 5, 5, 10, 15, 20, 25,
                 25, 24, 23, 22, 21, 21, 16, 11, 6, 1]

    def ext(out_edges, dest=True):
        res = []
        for o in out_edges:
            nt, npn = 'nt%d' % edge_maps[o], 'np%d' % o
            res.append('%s_%s' % ((nt, npn) if dest else (npn, nt)))
        return res

    srcs = [ext([12, 13, 14], False), ext([16, 18, 20], False),
            ext([2, 3, 4], False), ext([6, 8, 10], False)]
    sinks = [ext([2, 3, 4]), ext([6, 8, 10]), ext([14, 13, 12]), ext([20, 18, 16])]
    ratios1 = np.array([0.4, 0.7, 0.9, 1.0, 0.75, 0.5, 0.25])
    ratios2 = np.array([0.3, 0.8, 0.9, 1.0, 0.8, 0.6, 0.2])
    flows = [peak_flow1 * 0.6 * ratios1, peak_flow1 * ratios1,
             peak_flow2 * 0.6 * ratios2, peak_flow2 * ratios2]
    times = np.arange(0, 3001, 300)
    id1 = len(ratios1)
    id2 = len(times) - 1 - id1
    out = []
    for i in range(len(times) - 1):
        tb, te = int(times[i]), int(times[i + 1])
        if i < id1:
            for j in (0, 1):
                for e1, e2 in zip(srcs[j], sinks[j]):
                    out.append((e1, e2, tb, te, int(flows[j][i])))
        if i >= id2:
            for j in (2, 3):
                for e1, e2 in zip(srcs[j], sinks[j]):
                    out.append((e1, e2, tb, te, int(flows[j][i - id2])))
    return out


def large_grid_neighbor_map() -> Dict[str, List[str]]:
    """envs/large_grid_env.py:73-101 (internal nodes list N,E,S,W)."""
    m = {'nt1': ['nt6', 'nt2'], 'nt5': ['nt10', 'nt4'],
         'nt21': ['nt22', 'nt16'], 'nt25': ['nt20', 'nt24'],
         'nt2': ['nt7', 'nt3', 'nt1'], 'nt3': ['nt8', 'nt4', 'nt2'],
         'nt4': ['nt9', 'nt5', 'nt3'], 'nt22': ['nt23', 'nt17', 'nt21'],
         'nt23': ['nt24', 'nt18', 'nt22'], 'nt24': ['nt25', 'nt19', 'nt23'],
         'nt10': ['nt15', 'nt5', 'nt9'], 'nt15': ['nt20', 'nt10', 'nt14'],
         'nt20': ['nt25', 'nt15', 'nt19'], 'nt6': ['nt11', 'nt7', 'nt1'],
         'nt11': ['nt16', 'nt12', 'nt6'], 'nt16': ['nt21', 'nt17', 'nt11']}
    for i in (7, 8, 9, 12, 13, 14, 17, 18, 19):
        m['nt%d' % i] = ['nt%d' % (i + 5), 'nt%d' % (i + 1), 'nt%d' % (i - 5), 'nt%d' % (i - 1)]
    return m


# ---------------------------------------------------------------------------
# generic helpers
# ---------------------------------------------------------------------------
def _state_dims(agent, n_lane_ls, n_a_ls, neighbors, has_wait):
    """envs/env.py:303-323."""
    n_s, n_w, n_f = [], [], []
    for a, nl in enumerate(n_lane_ls):
        num_wave, num_fp = nl, 0
        for j in neighbors[a]:
            if agent not in ('a2c', 'greedy'):
                num_wave += n_lane_ls[j]
            if agent == 'ma2c':
                num_fp += n_a_ls[j] - 1
        num_wait = nl if has_wait else 0
        n_s.append(num_wave + num_wait + num_fp)
        n_w.append(num_wait)
        n_f.append(num_fp)
    return n_s, n_w, n_f


def _obs_table(agent, agent_lanes, agent_nlane, n_a_ls, neighbors, has_wait, a_max):
    """Gather table for envs/env.py:163-205: [own wave ; neighbour waves
    (*coop_gamma for ma2c) ; own wait ; neighbour fingerprints (ma2c)]."""
    rows = []
    for a in range(len(agent_nlane)):
        row = [(1, int(agent_lanes[a, k])) for k in range(agent_nlane[a])]
        if agent != 'greedy':
            if agent != 'a2c':
                for j in neighbors[a]:
                    kind = 2 if agent == 'ma2c' else 1
                    row += [(kind, int(agent_lanes[j, k])) for k in range(agent_nlane[j])]
            if has_wait:
                row += [(3, int(agent_lanes[a, k])) for k in range(agent_nlane[a])]
            if agent == 'ma2c':
                for j in neighbors[a]:
                    row += [(4, j * a_max + k) for k in range(n_a_ls[j] - 1)]
        rows.append(row)
    smax = max(len(r) for r in rows)
    smax = (smax + 3) // 4 * 4
    kind = np.zeros((len(rows), smax), np.int32)
    src = np.zeros((len(rows), smax), np.int32)
    for a, r in enumerate(rows):
        for j, (k, s) in enumerate(r):
            kind[a, j], src[a, j] = k, s
    return kind, src, [len(r) for r in rows]


def _signal_tables(phases_per_agent, kmax):
    A = len(phases_per_agent)
    pmax = max(len(p) for p in phases_per_agent)
    green = np.full((A, pmax, kmax), ord('r'), np.uint8)
    yellow = np.full((A, pmax, pmax, kmax), ord('r'), np.uint8)
    for a, ph in enumerate(phases_per_agent):
        for p, s in enumerate(ph):
            green[a, p, :len(s)] = np.frombuffer(s.encode(), np.uint8)
            for q, s0 in enumerate(ph):
                # yellow_tab[a, prev=q, new=p]
                y = s if q == p else yellow_phase(s0, s)
                yellow[a, q, p, :len(y)] = np.frombuffer(y.encode(), np.uint8)
    return green, yellow


# ---------------------------------------------------------------------------
# large_grid
# ---------------------------------------------------------------------------
_APPROACH = ('N', 'E', 'S', 'W')          # SUMO link order: incoming edges clockwise from north
_RIGHT, _THROUGH, _LEFT = 0, 1, 2


def four_leg_foes():
    """Foe links of a four-leg junction whose 12 signal links are ordered like netconvert orders them (and like
    LARGE_GRID_PHASES reads, envs/large_grid_env.py:40-41): legs clockwise from north, per leg right / through / left.
    Put every leg's incoming and outgoing side on a circle (clockwise: in, out; right-hand traffic): a movement is the
    chord from its leg's `in` to its target leg's `out`; two movements from different legs are foes when their chords cross
    or end on the same leg.  -> u32 [12] bit masks."""
    def chord(k):
        leg, turn = divmod(k, 3)
        out = (leg + (3, 2, 1)[turn]) % 4                # right / through / left
        return leg, out
    def between(a, b, x):                                # x strictly inside the clockwise arc a -> b
        return 0 < (x - a) % 8 < (b - a) % 8
    foes = np.zeros(12, np.uint32)
    for k in range(12):
        li, lo = chord(k)
        a, b = 2 * li, 2 * lo + 1
        for k2 in range(12):
            mi, mo = chord(k2)
            if mi == li:
                continue
            c, d = 2 * mi, 2 * mo + 1
            if mo == lo or (between(a, b, c) != between(a, b, d)):
                foes[k] |= np.uint32(1 << k2)
    return foes


def build_large_grid(agent: str = 'ma2c', peak_flow1: int = 1100, peak_flow2: int = 925,
                     sort_lanes: bool = True, init_density: float = 0.0, lane_change: bool = LANE_CHANGE_DEFAULT, **env_kw) -> Scenario:
    # lane_change (MICROSIM_SPEC.md rule 10): a hand-off enters the lane the junction's CONNECTION leads to (build_file.py:107-124:
    # through and right turns lane 0 -> lane 0, a left turn from an avenue -> street lane 1); a vehicle that then stands on the
    # street lane its next movement does not use has to move over to the sibling lane inside the edge.  False: the vehicle is put
    # on the lane its next movement needs right at edge entry (rounds 1 - 4)
    # init_density > 0: large_grid/data/build_file.py:223-266 seeds every internal edge (and both lanes of a street) with
    # int(30 * density) vehicles bound for a sink edge drawn per episode from np.random -- see the stream tables below
    L0, L0_END, N = 200.0, 75.0, 5
    pos: Dict[str, Tuple[float, float]] = {}
    for r in range(N):
        for c in range(N):
            pos['nt%d' % (1 + N * r + c)] = (L0 * c, L0 * r)
    for c in range(N):
        pos['np%d' % (1 + c)] = (L0 * c, -L0_END)
        pos['np%d' % (11 + (4 - c))] = (L0 * c, L0 * 4 + L0_END)
    for r in range(N):
        pos['np%d' % (6 + r)] = (L0 * 4 + L0_END, L0 * r)
        pos['np%d' % (16 + (4 - r))] = (-L0_END, L0 * r)

    def nbr(node, d):
        i = int(node[2:]) - 1
        r, c = divmod(i, N)
        if d == 'N':
            return 'nt%d' % (i + 1 + N) if r < N - 1 else 'np%d' % (11 + (4 - c))
        if d == 'S':
            return 'nt%d' % (i + 1 - N) if r > 0 else 'np%d' % (1 + c)
        if d == 'E':
            return 'nt%d' % (i + 2) if c < N - 1 else 'np%d' % (6 + r)
        return 'nt%d' % i if c > 0 else 'np%d' % (16 + (4 - r))

    tl_nodes = ['nt%d' % i for i in range(1, N * N + 1)]
    node_names = sorted(tl_nodes)                       # envs/env.py:232
    aidx = {n: i for i, n in enumerate(node_names)}

    # edges: id -> (from, to, nlanes, speed, length)
    edges: Dict[str, Tuple[str, str, int, float, float]] = {}
    for n in tl_nodes:
        for d in _APPROACH:
            o = nbr(n, d)
            street = d in ('E', 'W')
            nl, sp = (2, 20.0) if street else (1, 11.0)
            ln = L0 if o.startswith('nt') else L0_END
            edges['%s_%s' % (o, n)] = (o, n, nl, sp, ln)
            edges['%s_%s' % (n, o)] = (n, o, nl, sp, ln)

    # movement geometry at a TL node: approach a (edge arriving FROM direction a)
    # right/through/left -> leaving TOWARDS direction
    out_dir = {'N': ('W', 'S', 'E'), 'E': ('N', 'W', 'S'),
               'S': ('E', 'N', 'W'), 'W': ('S', 'E', 'N')}

    def in_edge(n, a):
        return '%s_%s' % (nbr(n, a), n)

    def out_edge(n, a, mv):
        return '%s_%s' % (n, nbr(n, out_dir[a][mv]))

    def approach_of(edge):
        fr, to = edges[edge][0], edges[edge][1]
        for a in _APPROACH:
            if nbr(to, a) == fr:
                return a
        raise KeyError(edge)

    # ---- routing: lexicographic (free-flow time, #turns), tie -> through, right, left
    demand = large_grid_demand(peak_flow1, peak_flow2)
    route_names: List[Tuple[str, str]] = []
    for e1, e2, *_ in demand:
        if (e1, e2) not in route_names:
            route_names.append((e1, e2))
    n_od = len(route_names)
    init_sinks: List[str] = []
    if init_density > 0:
        # routes are destination trees (Bellman-Ford below): one more per sink edge no OD flow ends at.  Sink order =
        # init_routes' sink_edges (build_file.py:227-236), which np.random.choice indexes
        in_nodes = [5, 10, 15, 20, 25, 21, 16, 11, 6, 1, 1, 2, 3, 4, 5, 25, 24, 23, 22, 21]
        out_nodes = [6, 7, 8, 9, 10, 16, 17, 18, 19, 20, 1, 2, 3, 4, 5, 11, 12, 13, 14, 15]
        init_sinks = ['nt%d_np%d' % (i, j) for i, j in zip(in_nodes, out_nodes)]
        for e in init_sinks:
            if all(dst != e for _, dst in route_names):
                route_names.append((None, e))
    NR = len(route_names)
    tcost = {e: Fraction(int(v[4] * 1000), int(v[3] * 1000)) for e, v in edges.items()}
    next_mv: Dict[Tuple[str, int], int] = {}
    for r, (_, dst) in enumerate(route_names):
        INF = (Fraction(10 ** 9), 10 ** 9)
        dist = {e: INF for e in edges}
        dist[dst] = (Fraction(0), 0)
        changed = True
        while changed:                                   # Bellman-Ford on 140 edges
            changed = False
            for e, v in edges.items():
                if e == dst or not v[1].startswith('nt'):
                    continue
                a = approach_of(e)
                best = dist[e]
                for mv in (_THROUGH, _RIGHT, _LEFT):
                    e2 = out_edge(v[1], a, mv)
                    d2 = dist[e2]
                    if d2 == INF:
                        continue
                    cand = (d2[0] + tcost[e2], d2[1] + (0 if mv == _THROUGH else 1))
                    if cand < best:
                        best = cand
                if best < dist[e]:
                    dist[e] = best
                    changed = True
        for e, v in edges.items():
            if e == dst or not v[1].startswith('nt') or dist[e] == INF:
                continue
            a = approach_of(e)
            for mv in (_THROUGH, _RIGHT, _LEFT):         # tie-break order
                e2 = out_edge(v[1], a, mv)
                if dist[e2] == INF:
                    continue
                if (dist[e2][0] + tcost[e2], dist[e2][1] + (0 if mv == _THROUGH else 1)) == dist[e]:
                    next_mv[(e, r)] = mv
                    break

    # ---- lanes: 6 incoming per agent in ild order [N_0,E_0,E_1,S_0,W_0,W_1], then exit lanes
    lane_names: List[str] = []
    lane_id: Dict[str, int] = {}
    agent_lanes = np.full((N * N, 6), -1, np.int32)
    link_lane = np.full((N * N, 12), -1, np.int32)
    for a_i, n in enumerate(node_names):
        k = 0
        for ai, a in enumerate(_APPROACH):
            e = in_edge(n, a)
            for ln in range(edges[e][2]):
                nm = '%s_%d' % (e, ln)
                lane_id[nm] = len(lane_names)
                lane_names.append(nm)
                agent_lanes[a_i, k] = lane_id[nm]
                k += 1
            # SUMO link order inside an approach: right, through (lane 0), left (leftmost lane)
            link_lane[a_i, 3 * ai + _RIGHT] = lane_id['%s_0' % e]
            link_lane[a_i, 3 * ai + _THROUGH] = lane_id['%s_0' % e]
            link_lane[a_i, 3 * ai + _LEFT] = lane_id['%s_%d' % (e, edges[e][2] - 1)]
    for e, v in edges.items():
        if v[1].startswith('np'):
            for ln in range(v[2]):
                nm = '%s_%d' % (e, ln)
                lane_id[nm] = len(lane_names)
                lane_names.append(nm)
    NL = len(lane_names)
    lane_edge = [nm.rsplit('_', 1)[0] for nm in lane_names]
    lane_k = [int(nm.rsplit('_', 1)[1]) for nm in lane_names]
    lane_len = np.array([edges[e][4] for e in lane_edge], np.float32)
    lane_vmax = np.array([edges[e][3] for e in lane_edge], np.float32)
    lane_node = np.array([aidx.get(edges[e][1], -1) for e in lane_edge], np.int32)
    lane_det = np.where(lane_node >= 0, lane_len - DET_LEN, 0).astype(np.float32)

    def lane_choice(edge, r, via_mv, from_street):
        """Lane a vehicle of route r takes when it enters `edge` (MICROSIM_SPEC.md
        'lane choice at edge entry').  On the arrival edge the lane follows the
        reference's connection table (build_file.py:107-124)."""
        if edges[edge][2] == 1:
            return 0
        if lane_change and via_mv is not None:          # the connection's lane, whatever the route needs at the far end
            return 1 if (via_mv == _LEFT and not from_street) else 0
        m = next_mv.get((edge, r))
        if m is None:                                   # arrival edge
            return 1 if (via_mv == _LEFT and not from_street) else 0
        return 1 if m == _LEFT else 0

    mv_next = np.full((NL, NR), -2, np.int32)
    mv_link = np.full((NL, NR), -1, np.int32)
    for l in range(NL):
        e = lane_edge[l]
        fr, to, nl, _, _ = edges[e]
        for r, (_, dst) in enumerate(route_names):
            if e == dst:
                mv_next[l, r] = -1
                continue
            if init_density > 0 and to.startswith('np'):
                mv_next[l, r] = -1                      # initial traffic: whoever ends up on an exit edge leaves there
                continue
            m = next_mv.get((e, r))
            if m is None or not to.startswith('nt'):
                continue
            if nl == 2 and ((m == _LEFT) != (lane_k[l] == 1)):
                if init_density <= 0 or lane_change:
                    continue                            # lane does not serve that movement (lane_change: the vehicle moves over)
                # initial traffic stands on both lanes of a street whatever its sink (departLane, build_file.py:225);
                # SUMO would change lanes, here the vehicle takes a movement its lane serves -- the left turn from lane 1,
                # through (else right) from lane 0 -- and follows the sink's tree from the next edge on
                m = _LEFT if lane_k[l] == 1 else _THROUGH
            a = approach_of(e)
            e2 = out_edge(to, a, m)
            k2 = lane_choice(e2, r, m, nl == 2)
            mv_next[l, r] = lane_id['%s_%d' % (e2, k2)]
            mv_link[l, r] = 3 * _APPROACH.index(a) + m
    route_entry = np.array([lane_id['%s_%d' % (src, lane_choice(src, r, None, False))] if src is not None else -1
                            for r, (src, _) in enumerate(route_names)], np.int32)

    lane_sib = None
    if lane_change:
        lane_sib = np.array([lane_id.get('%s_%d' % (lane_edge[l], 1 - lane_k[l]), -1) if edges[lane_edge[l]][2] == 2 else -1
                             for l in range(NL)], np.int32)
    lane_up = np.full((NL, MAX_UP), -1, np.int32)
    for l2 in range(NL):
        ups = sorted({l for l in range(NL) if (mv_next[l] == l2).any()})
        if lane_sib is not None and lane_sib[l2] >= 0:
            ups = [int(lane_sib[l2])] + ups             # lane changers keep their position: gathered before the junction's arrivals
        assert len(ups) <= MAX_UP
        lane_up[l2, :len(ups)] = ups

    # right of way (MICROSIM_SPEC.md, rule 2): left turns yield to the opposing approach's lane-0 head
    # when that head goes right / through -- the only merge conflict the five phases admit
    opp = {'N': 'S', 'S': 'N', 'E': 'W', 'W': 'E'}
    mv_yield = np.full((NL, NR), -1, np.int32)
    mv_prio = np.zeros((NL, NR), np.int32)
    for l in range(NL):
        e = lane_edge[l]
        if lane_node[l] < 0:
            continue
        to = edges[e][1]
        for r in range(NR):
            if mv_link[l, r] < 0:
                continue
            if mv_link[l, r] % 3 == _LEFT:
                mv_yield[l, r] = lane_id['%s_0' % in_edge(to, opp[approach_of(e)])]
            else:
                mv_prio[l, r] = 1

    nmap = large_grid_neighbor_map()
    neighbors = [[aidx[j] for j in nmap[n]] for n in node_names]
    n_a_ls = [len(LARGE_GRID_PHASES)] * (N * N)
    nlane = [6] * (N * N)
    n_s, n_w, n_f = _state_dims(agent, nlane, n_a_ls, neighbors, True)
    obs_kind, obs_src, lens = _obs_table(agent, agent_lanes, nlane, n_a_ls, neighbors, True, 5)
    if agent not in ('greedy', 'a2c'):
        assert lens == n_s
    green, yellow = _signal_tables([LARGE_GRID_PHASES] * (N * N), 12)

    rid = {rn: i for i, rn in enumerate(route_names)}
    flows = np.array([[tb, te, vph, rid[(e1, e2)]] for e1, e2, tb, te, vph in demand], np.int32)
    stream_kw = {}
    if init_density > 0:
        # streams: the n_od flows of the demand table, then init_routes' 120 flows in its own order (build_file.py:243-266:
        # streets nt(i+j) <-> nt(i+j+1), both directions x both lanes; avenues nt(i+j) <-> nt(i+j+5)), each
        # `number = int(MAX_CAR_NUM * density)` vehicles at t = 0, sink = np.random.choice(sink_edges) per episode
        car_num = int(30 * init_density)
        dst_route = {dst: r for r, (_, dst) in enumerate(route_names)}
        sink_routes = [dst_route[e] for e in init_sinks]
        init_lanes = []
        for i in range(1, 25, 5):
            for j in range(4):
                n1, n2 = 'nt%d' % (i + j), 'nt%d' % (i + j + 1)
                init_lanes += ['%s_%s_0' % (n1, n2), '%s_%s_0' % (n2, n1), '%s_%s_1' % (n1, n2), '%s_%s_1' % (n2, n1)]
        for i in range(1, 6):
            for j in range(0, 20, 5):
                n1, n2 = 'nt%d' % (i + j), 'nt%d' % (i + j + 5)
                init_lanes += ['%s_%s_0' % (n1, n2), '%s_%s_0' % (n2, n1)]
        NS = n_od + len(init_lanes)
        KC = len(sink_routes)
        choice = np.full((NS, 1, KC, 2), -1, np.int32)
        choice[..., 1] = 0
        for r in range(n_od):
            choice[r, 0, 0] = (r, 65536)
        for k in range(len(init_lanes)):
            for c, r in enumerate(sink_routes):
                choice[n_od + k, 0, c] = (r, 65536 * (c + 1) // KC)
        stream_kw = dict(stream_entry_lane=np.array(list(route_entry[:n_od]) + [lane_id[nm] for nm in init_lanes], np.int32),
                         stream_mode=np.array([0] * n_od + [2] * len(init_lanes), np.int32), stream_choice=choice)
        if car_num > 0:
            flows = np.concatenate([flows, np.array([[0, 1, 3600 * car_num, n_od + k] for k in range(len(init_lanes))], np.int32)])
        route_entry = np.where(route_entry >= 0, route_entry, 0).astype(np.int32)

    scn = Scenario(
        name='large_grid', agent=agent, node_names=node_names, n_agent=N * N,
        lane_names=lane_names, lane_len=lane_len, lane_vmax=lane_vmax, lane_node=lane_node,
        lane_det_start=lane_det, lane_up=lane_up, lane_sib=lane_sib,
        n_route=NR, mv_next=mv_next, mv_link=mv_link, mv_yield=mv_yield, mv_prio=mv_prio,
        mv_zip=np.zeros((NL, NR), np.int32), route_entry_lane=route_entry,
        route_names=route_names,
        agent_lanes=agent_lanes, agent_nlane=np.array(nlane, np.int32),
        agent_nlink=np.full(N * N, 12, np.int32), agent_nphase=np.array(n_a_ls, np.int32),
        link_lane=link_lane, phases=[LARGE_GRID_PHASES] * (N * N), link_foes=np.tile(four_leg_foes(), (N * N, 1)),
        green_tab=green, yellow_tab=yellow,
        neighbors=neighbors, n_s_ls=n_s, n_w_ls=n_w, n_f_ls=n_f, n_a_ls=n_a_ls,
        obs_kind=obs_kind, obs_src=obs_src, flows=flows, obs_len=lens,
        extra={'peak_flow1': peak_flow1, 'peak_flow2': peak_flow2, 'demand': demand, 'lane_change': bool(lane_change)},
        **stream_kw, **env_kw)
    if init_density > 0:
        scn.extra.update(init_density=init_density, init_lanes=init_lanes, init_sinks=init_sinks, sink_routes=sink_routes,
                         n_od=n_od, car_num=car_num)
    return sort_lanes_by_load(scn) if sort_lanes else scn


def draw_stream_routes(scn: Scenario, seed: int):
    """This episode's routes of the streams whose route the reference's generator draws (mode 2): int32 [NS] or None.
    large_grid with init_density > 0: gen_rou_file(seed) does np.random.seed(seed) and then one
    np.random.choice(sink_edges) per initial flow, in flow order (large_grid/data/build_file.py:237-240,275-281);
    RandomState(seed).choice(20) is that stream (pinned against the generator's file by tests/test_init_density.py)."""
    if scn.stream_mode is None or not (np.asarray(scn.stream_mode) == 2).any():
        return None
    rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
    routes = scn.stream_choice[:, 0, 0, 0].astype(np.int32).copy()
    m2 = np.nonzero(np.asarray(scn.stream_mode) == 2)[0]
    cand = scn.stream_choice[m2, 0, :, 0]                      # [n, KC] candidate routes of every drawn stream, -1 padded
    K = (cand >= 0).sum(axis=1)
    if (K == K[0]).all():
        # one vectorised draw: RandomState.choice(K, size=n) consumes the legacy stream exactly like n scalar choice(K)
        # calls (tests/test_init_density.py), at 1/1000 of their cost (E x 120 calls per reset before)
        routes[m2] = cand[np.arange(len(m2)), rs.choice(int(K[0]), size=len(m2))]
    else:
        for j, s_ in enumerate(m2):
            routes[s_] = cand[j, int(rs.choice(int(K[j])))]
    return routes



# ---------------------------------------------------------------------------
# real_net (Monaco)
# ---------------------------------------------------------------------------
def build_real_net(agent: str = 'ma2c', flow_rate: int = 325, sort_lanes: bool = True, contract: bool = True,
                   **env_kw) -> Scenario:
    """Monaco scenario (envs/real_net_env.py) from the compiled table file
    ``data/real_net.json`` (tools/compile_real_net.py: most.net.xml lanes / connections / signal links,
    NODES + PHASES of envs/real_net_env.py:20-68, flows of real_net/data/build_file.py:27-104).

    Reference semantics kept: agent order = sorted node ids (env.py:232); a node's lanes = dedup of
    its signal links' incoming lanes in link order (env.py:219-230); wave-only state measured on
    the whole lane (env.py:376-377, real_net_env.py:18); reward = -sum(min(10, halting))
    (env.py:332-333, objective queue); rewards / ((1+deg)*20) (env.py:599-601,625-629);
    --time-to-teleport 300 (env.py:283-284); every active flow at `flow_rate` veh/h.
    Microsim-side choices (MICROSIM_SPEC.md): zero-length junctions, unsignalised junctions always open,
    free lane choice at edge entry among the lanes that continue the route, no right-of-way table."""
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'real_net.json')))
    node_names = sorted(d['nodes'])
    aidx = {n: i for i, n in enumerate(node_names)}
    A = len(node_names)
    controlled = {}                                            # lane name -> agent
    for n in node_names:
        for lane in d['tl_links'][n].values():
            controlled[lane] = aidx[n]
    lane_names, lane_id = [], {}
    for e in sorted(d['edges']):
        for i, (ln, sp, ok) in enumerate(d['edges'][e]['lanes']):
            nm = '%s_%d' % (e, i)
            if ok or nm in controlled:
                lane_id[nm] = len(lane_names)
                lane_names.append(nm)
    NL = len(lane_names)
    lane_edge = [nm.rsplit('_', 1)[0] for nm in lane_names]
    lane_k = [int(nm.rsplit('_', 1)[1]) for nm in lane_names]
    lane_len = np.array([d['edges'][e]['lanes'][k][0] for e, k in zip(lane_edge, lane_k)], np.float32)
    lane_vmax = np.array([d['edges'][e]['lanes'][k][1] for e, k in zip(lane_edge, lane_k)], np.float32)
    lane_node = np.array([controlled.get(nm, -1) for nm in lane_names], np.int32)
    conn = {}                                                   # (edge, lane) -> {to_edge: [(to_lane, node, link)]}
    for f, fl, t, tl_, node, link in d['connections']:
        conn.setdefault((f, fl), {}).setdefault(t, []).append((tl_, node, link))
    routes = d['routes']
    NR = len(routes)

    def continues(e, k, nxt):
        return nxt is None or nxt in conn.get((e, k), {})

    mv_next = np.full((NL, NR), -2, np.int32)
    mv_link = np.full((NL, NR), -1, np.int32)
    route_entry = np.zeros(NR, np.int32)
    for r, path in enumerate(routes):
        assert len(set(path)) == len(path), 'route %d visits an edge twice' % r
        for p, e in enumerate(path):
            e2 = path[p + 1] if p + 1 < len(path) else None
            e3 = path[p + 2] if p + 2 < len(path) else None
            for k in range(len(d['edges'][e]['lanes'])):
                nm = '%s_%d' % (e, k)
                if nm not in lane_id:
                    continue
                l = lane_id[nm]
                if e2 is None:
                    mv_next[l, r] = -1
                    continue
                cs = sorted(conn.get((e, k), {}).get(e2, []))
                if not cs:
                    continue
                good = [c for c in cs if continues(e2, c[0], e3)]
                if good:
                    tk, node, link = good[0]
                else:                                           # free lane choice at edge entry
                    cand = [k2 for k2 in range(len(d['edges'][e2]['lanes']))
                            if '%s_%d' % (e2, k2) in lane_id and d['edges'][e2]['lanes'][k2][2] and continues(e2, k2, e3)]
                    assert cand, 'route %d is not drivable at %s -> %s' % (r, e, e2)
                    tk, (node, link) = cand[0], cs[0][1:]
                mv_next[l, r] = lane_id['%s_%d' % (e2, tk)]
                mv_link[l, r] = link if node in aidx else -1
        first = [k for k in range(len(d['edges'][path[0]]['lanes']))
                 if '%s_%d' % (path[0], k) in lane_id and d['edges'][path[0]]['lanes'][k][2] and continues(path[0], k, path[1])]
        route_entry[r] = lane_id['%s_%d' % (path[0], first[0])]
        l, hops = int(route_entry[r]), 0                        # every route must reach its last edge
        while mv_next[l, r] >= 0:
            l, hops = int(mv_next[l, r]), hops + 1
        assert mv_next[l, r] == -1 and hops == len(path) - 1, 'route %d broken' % r
    lane_up = np.full((NL, MAX_UP), -1, np.int32)
    for l2 in range(NL):
        ups = sorted({l for l in range(NL) if (mv_next[l] == l2).any()})
        assert len(ups) <= MAX_UP, 'lane %s has %d feeders' % (lane_names[l2], len(ups))
        lane_up[l2, :len(ups)] = ups
    # zipper merge (MICROSIM_SPEC.md): a lane with several feeders takes arrivals from feeder `rank`
    # only in seconds with (t + rank) % count == 0
    mv_zip = np.zeros((NL, NR), np.int32)
    for l in range(NL):
        for r in range(NR):
            t2 = mv_next[l, r]
            if t2 >= 0:
                ups = [u for u in lane_up[t2] if u >= 0]
                if len(ups) > 1:
                    mv_zip[l, r] = ups.index(l) | (len(ups) << 8)
    phases = [d['phases'][d['nodes'][n]['phase']] for n in node_names]
    kmax = max(len(p[0]) for p in phases)
    lmax = 0
    link_lane = np.full((A, kmax), -1, np.int32)
    lanes_per_agent = []
    for a, n in enumerate(node_names):
        links = d['tl_links'][n]
        assert len(links) == len(phases[a][0])                  # env.py:159
        seq = [lane_id[links[str(k)]] for k in range(len(links))]
        link_lane[a, :len(seq)] = seq
        ded = []
        for l in seq:
            if l not in ded:
                ded.append(l)
        lanes_per_agent.append(ded)
        lmax = max(lmax, len(ded))
    agent_lanes = np.full((A, lmax), -1, np.int32)
    for a, ded in enumerate(lanes_per_agent):
        agent_lanes[a, :len(ded)] = ded
    nlane = [len(x) for x in lanes_per_agent]
    neighbors = [[aidx[j] for j in d['nodes'][n]['neighbors']] for n in node_names]
    n_a_ls = [len(p) for p in phases]
    a_max = max(n_a_ls)
    n_s, n_w, n_f = _state_dims(agent, nlane, n_a_ls, neighbors, False)
    obs_kind, obs_src, lens = _obs_table(agent, agent_lanes, nlane, n_a_ls, neighbors, False, a_max)
    if agent not in ('greedy', 'a2c'):
        assert lens == n_s
    green, yellow = _signal_tables(phases, kmax)
    link_foes = np.zeros((A, kmax), np.uint32)              # the junctions' own right-of-way matrices (tools/compile_real_net.py)
    for a, n in enumerate(node_names):
        for k, fo in d.get('foes', {}).get(n, {}).items():
            for k2 in fo:
                link_foes[a, int(k)] |= np.uint32(1 << int(k2))
    flows = np.array([[b, e, flow_rate, r] for r, b, e in d['flows']], np.int32)
    kw = dict(objective='queue', coef_wait=0.0, norm_wait=30.0, has_wait_state=False, queue_cap=10,
              reward_scale_realnet=True, teleport_sec=300)
    kw.update(env_kw)
    scn = Scenario(
        name='real_net', agent=agent, node_names=node_names, n_agent=A,
        lane_names=lane_names, lane_len=lane_len, lane_vmax=lane_vmax, lane_node=lane_node,
        lane_det_start=np.zeros(NL, np.float32), lane_up=lane_up,
        n_route=NR, mv_next=mv_next, mv_link=mv_link, mv_yield=np.full((NL, NR), -1, np.int32),
        mv_prio=np.zeros((NL, NR), np.int32), mv_zip=mv_zip, route_entry_lane=route_entry,
        route_names=[(p[0], p[-1]) for p in routes],
        agent_lanes=agent_lanes, agent_nlane=np.array(nlane, np.int32),
        agent_nlink=np.array([len(p[0]) for p in phases], np.int32), agent_nphase=np.array(n_a_ls, np.int32),
        link_lane=link_lane, phases=phases, green_tab=green, yellow_tab=yellow, link_foes=link_foes,
        neighbors=neighbors, n_s_ls=n_s, n_w_ls=n_w, n_f_ls=n_f, n_a_ls=n_a_ls,
        obs_kind=obs_kind, obs_src=obs_src, flows=flows, obs_len=lens,
        extra={'flow_rate': flow_rate, 'routes': routes}, **kw)
    if contract:
        scn = contract_chains(scn)
    return sort_lanes_by_load(scn) if sort_lanes else scn


def contract_chains(scn: Scenario) -> Scenario:
    """Merge 1-to-1 lane chains across unsignalised junctions (MICROSIM_SPEC.md, "lane chains").

    SUMO splits Monaco's roads at every geometry node: half of the lanes a route uses are 2-30 m pieces in series.  With
    zero-length junctions, one lane hop per second and the room rule at lane entry such a chain is a far worse
    bottleneck than the road it models.  Lane A is merged into its successor B when B is A's only successor on every
    route, A is B's only feeder and the junction between them is unsignalised: the merged lane keeps B's name, signal,
    movements and speed limit, is len(A) + len(B) long, and B's detector (whole lane, envs/env.py:376-377) becomes
    x >= len(A) -- observations and rewards still count exactly the vehicles on the original lane B."""
    NL, NR = scn.n_lane, scn.n_route
    nxt = [set(int(x) for x in scn.mv_next[l] if x >= 0) for l in range(NL)]
    ends = [bool((scn.mv_next[l] == -1).any()) for l in range(NL)]
    ups = [[int(u) for u in scn.lane_up[l] if u >= 0] for l in range(NL)]
    cand = {}                                   # A -> B mergeable pairs
    for a in range(NL):
        if scn.lane_node[a] >= 0 or ends[a] or len(nxt[a]) != 1:
            continue
        b = next(iter(nxt[a]))
        # a movement of A that yields or has priority at the junction A -> B is a right-of-way relation the merged lane
        # would silently drop: such a junction stays (none on the reference's Monaco routes, asserted by
        # tests/test_scenario_tables.py)
        if (scn.mv_yield[a] >= 0).any() or (scn.mv_prio[a] != 0).any():
            continue
        if ups[b] == [a] and b != a:
            cand[a] = b
    # a merged lane must still fit its vehicles into LANE_CAP slots: grow chains from their downstream end while the
    # total stays within what LANE_CAP - MAX_CROSS standing vehicles occupy; where a chain is cut the walk starts anew
    max_len = (LANE_CAP - MAX_CROSS) * (VEH_LEN + MIN_GAP)
    into = {}                                   # A -> B
    work = [l for l in range(NL) if l not in cand]
    while work:
        cur = work.pop()
        total = float(scn.lane_len[cur])
        while len(ups[cur]) == 1 and cand.get(ups[cur][0]) == cur:
            a = ups[cur][0]
            if total + float(scn.lane_len[a]) > max_len:
                work.append(a)
                break
            into[a] = cur
            total += float(scn.lane_len[a])
            cur = a
    if not into:
        return scn
    head_of = {}                                # every lane -> (last lane of its chain, offset of its start in the merged lane)
    def resolve(l):
        chain = [l]
        while chain[-1] in into:
            chain.append(into[chain[-1]])
        return chain
    keep = [l for l in range(NL) if l not in into]
    new_id = {l: i for i, l in enumerate(keep)}
    first_of = {l: l for l in keep}             # kept lane -> first lane of its chain
    offset = {l: 0.0 for l in keep}
    for a in sorted(into):
        if any(into.get(u) == a for u in range(NL)):
            continue                            # not the first lane of its chain
        chain = resolve(a)
        last = chain[-1]
        first_of[last] = a
        offset[last] = float(sum(np.float32(scn.lane_len[c]) for c in chain[:-1]))
    def remap_lane(v):
        v = np.asarray(v)
        out = v.copy()
        for idx in np.ndindex(v.shape):
            if v[idx] >= 0:
                out[idx] = new_id[resolve(int(v[idx]))[-1]]
        return out
    old_len = scn.lane_len.copy()
    lane_len = np.array([np.float32(offset[l]) + scn.lane_len[l] for l in keep], np.float32)
    det = np.array([np.float32(offset[l]) + scn.lane_det_start[l] for l in keep], np.float32)
    lane_up = np.full((len(keep), MAX_UP), -1, np.int32)
    for i, l in enumerate(keep):
        fs = sorted({new_id[resolve(u)[-1]] for u in ups[first_of[l]]})
        lane_up[i, :len(fs)] = fs
    mv_next = remap_lane(scn.mv_next[keep])
    # a route that entered the chain at an inner lane keeps its movement rows from the last lane; rows of the dropped
    # lanes only said "go to the next piece"
    scn.extra['contracted'] = {scn.lane_names[a]: scn.lane_names[resolve(a)[-1]] for a in into}
    scn.lane_names = [scn.lane_names[l] for l in keep]
    scn.lane_len, scn.lane_det_start = lane_len, det
    scn.lane_origin = np.array([np.float32(offset[l]) + scn.lane_origin[l] for l in keep], np.float32)
    scn.lane_vmax, scn.lane_node = scn.lane_vmax[keep], scn.lane_node[keep]
    scn.lane_up, scn.mv_next = lane_up, mv_next.astype(np.int32)
    scn.mv_link, scn.mv_prio = scn.mv_link[keep], scn.mv_prio[keep]
    scn.mv_yield = remap_lane(scn.mv_yield[keep]).astype(np.int32)
    zp = np.zeros_like(scn.mv_zip[keep])
    for i in range(len(keep)):                  # zipper slots follow the new feeder lists
        for r in range(NR):
            t2 = scn.mv_next[i, r]
            if t2 >= 0:
                fs = [u for u in scn.lane_up[t2] if u >= 0]
                if len(fs) > 1:
                    zp[i, r] = fs.index(i) | (len(fs) << 8)
    scn.mv_zip = zp
    # a route whose SUMO entry lane became an inner piece of a chain is still inserted on that piece: the insertion window
    # of its stream starts where the piece does (include/tsc.h stream_origin)
    entry_off, entry_end = np.zeros(NR, np.float32), np.zeros(NR, np.float32)
    for r in range(NR):
        x = int(scn.route_entry_lane[r])
        full = resolve(first_of[resolve(x)[-1]])
        entry_off[r] = np.float32(sum(np.float32(scn.lane_len[c]) for c in full[:full.index(x)]))
        entry_end[r] = entry_off[r] + np.float32(old_len[x])
    scn.route_entry_lane = remap_lane(scn.route_entry_lane).astype(np.int32)
    if entry_off.any() or (entry_end < scn.lane_len[scn.route_entry_lane]).any():
        scn.stream_entry_lane = scn.route_entry_lane.copy()
        scn.stream_origin, scn.stream_limit = entry_off, entry_end
        scn.stream_mode = np.zeros(NR, np.int32)
        scn.stream_choice = np.stack([np.arange(NR), np.full(NR, 65536)], 1).reshape(NR, 1, 1, 2).astype(np.int32)
    scn.agent_lanes = remap_lane(scn.agent_lanes).astype(np.int32)
    scn.link_lane = remap_lane(scn.link_lane).astype(np.int32)
    src = scn.obs_src.copy()
    m = (scn.obs_kind >= 1) & (scn.obs_kind <= 3)
    src[m] = remap_lane(scn.obs_src[m])
    scn.obs_src = src.astype(np.int32)
    return scn


# ---------------------------------------------------------------------------
# small_grid (6 intersections)
# ---------------------------------------------------------------------------
SMALL_GRID_NEIGHBOR_MAP = {'nt1': ['npc', 'nt2', 'nt6'], 'nt2': ['nt1', 'nt3'], 'nt3': ['npc', 'nt2', 'nt4'],
                           'nt4': ['nt3', 'nt5'], 'nt5': ['npc', 'nt4', 'nt6'], 'nt6': ['nt1', 'nt5']}    # envs/small_grid_env.py:20-25
SMALL_GRID_STATE_PHASE_MAP = {'nt1': [0, 1, 2], 'nt2': [1, 0], 'nt3': [1, 0], 'nt4': [1, 0], 'nt5': [1, 0], 'nt6': [1, 0]}   # :29-30
SMALL_GRID_PHASES = {2: ['GGrr', 'rrGG'], 3: ['GGGrrrrrr', 'rrrGGGrrr', 'rrrrrrGGG']}                                      # :35-36


def small_grid_demand(num_car_hourly: int, episode_length_sec: int = 3600, distributions: bool = False):
    """The demand of small_grid/data/build_file.py as (edge path, begin, end, vph) elements.

    distributions=True returns instead (source flows, mf flows): the source flows as (source edge, interval, begin, end, vph,
    [(edge path, probability)]) -- what SUMO's jtrrouter samples every vehicle's turns from (build_small_grid draws them per
    vehicle, `random_turns`) -- and the explicit-route flows as (edge path, begin, end, vph).

    The reference writes source flows (`from=` only, vehsPerHour per 600 s, :191-210) and lets SUMO's `jtrrouter` draw every
    vehicle's turns from the ratios of output_turns (:223-307) with a per-episode seed; `jtrrouter` is not in this
    image.  Here a source flow is split over its possible paths in proportion to the product of the turn ratios
    (largest remainders, the total is kept): the expectation of the random routing, the same for every seed.  The
    "mf_" flows (`probability=` num_car_hourly / 3600 per second on explicit routes, :167-190) become
    vehsPerHour = round(3600 * that two-decimal probability)."""
    src_flows = [[500, 100, 700, 800, 550, 550, 100, 200, 250, 250, 400, 800],
                 [600, 700, 100, 200, 50, 100, 1000, 500, 450, 150, 400, 200],
                 [100, 400, 400, 200, 600, 550, 100, 500, 500, 800, 400, 200],
                 [100, 200, 300, 300, 300, 400, 600, 600, 800, 500, 400, 300],
                 [600, 400, 400, 600, 800, 400, 300, 300, 300, 200, 250, 250]]
    srcs = ['np1_nt1', 'np2_nt1', 'np3_nt1', 'np8_nt4', 'np9_nt4']
    turns = {'np1_nt1': {'nt1_nt2': 0.2, 'nt1_nt6': 0.5, 'nt1_npc': 0.3},
             'np2_nt1': {'nt1_nt2': 0.15, 'nt1_nt6': 0.15, 'nt1_npc': 0.7},
             'np3_nt1': {'nt1_nt2': 0.5, 'nt1_nt6': 0.15, 'nt1_npc': 0.35},
             'np8_nt4': {'nt4_nt3': 0.4, 'nt4_nt5': 0.6}, 'np9_nt4': {'nt4_nt3': 0.6, 'nt4_nt5': 0.4},
             'nt3_nt2': {'nt2_np5': 1.0}, 'nt1_nt2': {'nt2_np4': 1.0}, 'nt5_nt6': {'nt6_np12': 1.0}, 'nt1_nt6': {'nt6_np13': 1.0},
             'npc_nt3': {'nt3_nt2': 0.3, 'nt3_np6': 0.7}, 'npc_nt5': {'nt5_nt6': 0.3, 'nt5_np11': 0.7},
             # nt4_nt3 / nt4_nt5 have no <fromEdge> entry in the reference's turn file: jtrrouter then splits uniformly
             'nt4_nt3': {'nt3_nt2': 0.5, 'nt3_np6': 0.5}, 'nt4_nt5': {'nt5_nt6': 0.5, 'nt5_np11': 0.5}}
    base_probs = np.array([[0.15, 0.15], [0.35, 0.35], [0.15, 0.2]])
    sinks = {'nt6_np12', 'nt6_np13', 'nt2_np4', 'nt2_np5', 'nt5_np11', 'nt3_np6'}

    def paths(edge, i0):
        if edge in sinks:
            return [([edge], 1.0)]
        if edge == 'nt1_npc':                               # time-variant ratios at npc (:283-293)
            cur = np.ravel(np.dot(np.array(src_flows[:3])[:, i0].reshape(1, 3), base_probs))
            cur = cur / np.sum(cur)
            t = {'npc_nt3': float(cur[0]), 'npc_nt5': float(cur[1])}
        else:
            t = turns[edge]
        out = []
        for e2, p in t.items():
            out += [([edge] + q, p * pq) for q, pq in paths(e2, i0)]
        return out
    elements, src_elems = [], []
    for i0 in range(12):
        tb, te = 600 * i0, 600 * (i0 + 1)
        if tb >= episode_length_sec:
            break
        for j, src in enumerate(srcs):
            vph = int(src_flows[j][i0] * 1.0)                # FLOW_MULTIPLIER = 1.0, "%i"
            pp = paths(src, i0)
            src_elems.append((src, i0, tb, te, vph, [(tuple(path), float(pr)) for path, pr in pp]))
            raw = [vph * p for _, p in pp]
            alloc = [int(np.floor(x)) for x in raw]
            for k in sorted(range(len(pp)), key=lambda k: (-(raw[k] - alloc[k]), k))[:vph - sum(alloc)]:
                alloc[k] += 1
            elements += [(tuple(path), tb, te, v) for (path, _), v in zip(pp, alloc) if v > 0]
    mf_routes = ['nt1_npc npc_nt5 nt5_np11', 'nt1_npc npc_nt5 nt5_nt6 nt6_np12', 'nt4_nt5 nt5_np11', 'nt4_nt5 nt5_nt6 nt6_np12',
                 'nt1_nt2 nt2_np4', 'nt1_nt6 nt6_np13', 'nt1_npc npc_nt3 nt3_np6', 'nt1_npc npc_nt3 nt3_nt2 nt2_np5',
                 'nt4_nt3 nt3_np6', 'nt4_nt3 nt3_nt2 nt2_np5']
    cases = [(3, 4, 5), (0, 3, 4), (1, 2, 5), (4, 5, 9), (5, 6, 9), (4, 7, 8)]
    mf_vph = int(round(3600 * float('%.2f' % (num_car_hourly / float(3600)))))
    mf_elems = []
    for i, cs in enumerate(cases):
        tb, te = 1200 * i, 1200 * (i + 1)
        if tb >= episode_length_sec or mf_vph <= 0:
            continue
        mf_elems += [(tuple(mf_routes[c].split()), tb, te, mf_vph) for c in cs]
    if distributions:
        return src_elems, mf_elems
    return elements + mf_elems


def build_small_grid(agent: str = 'greedy', num_extra_car_per_hour: int = 1000, sort_lanes: bool = True,
                     random_turns: bool = True, **env_kw) -> Scenario:
    """The 6-intersection benchmark (envs/small_grid_env.py, small_grid/data/build_file.py): 1-lane 20 m/s roads, nt1 with
    three 1-approach phases, the others with two, lane-area detectors on the last 50 m of every incoming lane, the
    unsignalised split node npc.  SUMO orders a node's signal links by incoming edge clockwise from north; the
    hard-coded greedy map SMALL_GRID_STATE_PHASE_MAP is restated as is.  MARL agents: the reference's neighbour map lists
    the non-signal node 'npc' and crashes; it is dropped from the neighbour lists here."""
    L0, L1, LE = 200.0, 400.0, 75.0
    L2, L2E = L0 / np.sqrt(2), LE / np.sqrt(2)
    pos = {'nt1': (0, 0), 'nt2': (L1, 0), 'nt3': (L1, L0), 'nt4': (L1, L1), 'nt5': (L0, L1), 'nt6': (0, L1),
           'np1': (0, -LE), 'np2': (-L2E, -L2E), 'np3': (-LE, 0), 'np4': (LE + L1, 0), 'np5': (L1, -LE), 'np6': (LE + L1, L0),
           'np8': (LE + L1, L1), 'np9': (L1, LE + L1), 'np11': (L0, LE + L1), 'np12': (-LE, L1), 'np13': (0, LE + L1),
           'npc': (L2, L2)}
    edge_list = [('np1', 'nt1'), ('np2', 'nt1'), ('np3', 'nt1'), ('np8', 'nt4'), ('np9', 'nt4'), ('nt1', 'nt2'), ('nt1', 'npc'),
                 ('nt1', 'nt6'), ('npc', 'nt3'), ('npc', 'nt5'), ('nt5', 'nt6'), ('nt4', 'nt3'), ('nt4', 'nt5'), ('nt3', 'nt2'),
                 ('nt6', 'np12'), ('nt6', 'np13'), ('nt2', 'np4'), ('nt2', 'np5'), ('nt5', 'np11'), ('nt3', 'np6')]
    elen = {'%s_%s' % e: float(np.hypot(pos[e[1]][0] - pos[e[0]][0], pos[e[1]][1] - pos[e[0]][1])) for e in edge_list}
    conns = {}                                                    # in-edge -> out-edges (build_file.py:107-152)
    for e in edge_list:
        conns['%s_%s' % e] = ['%s_%s' % o for o in edge_list if o[0] == e[1] and o[1] != e[0]]
    node_names = sorted(['nt%d' % i for i in range(1, 7)])
    aidx = {n: i for i, n in enumerate(node_names)}

    def bearing(n, frm):                                          # clockwise from north, of the node an edge comes from
        dx, dy = pos[frm][0] - pos[n][0], pos[frm][1] - pos[n][1]
        return (np.degrees(np.arctan2(dx, dy)) + 360.0) % 360.0
    # a lane longer than LANE_CAP standing vehicles is cut into equal pieces joined 1-to-1 (the last piece keeps the
    # edge's name: it carries the signal and the detector); 200 m lanes stay whole, as in large_grid
    max_len = LANE_CAP * (VEH_LEN + MIN_GAP)
    lane_names, lane_len, piece_first, piece_last = [], [], {}, {}
    for e in ['%s_%s' % x for x in edge_list]:
        k = int(np.ceil(elen[e] / max_len))
        names = ['%s_0#%d' % (e, i) for i in range(k - 1)] + ['%s_0' % e]
        piece_first[e], piece_last[e] = len(lane_names), len(lane_names) + k - 1
        lane_names += names
        lane_len += [elen[e] / k] * k
    NL = len(lane_names)
    lane_len = np.array(lane_len, np.float32)
    # signal links: per TL node, incoming edges clockwise from north, their out-edges in the same sense
    link_edges, phases = {}, []
    for n in node_names:
        ins = sorted([e for e in edge_list if e[1] == n], key=lambda e: bearing(n, e[0]))
        links = []
        for e in ins:
            outs = sorted(conns['%s_%s' % e], key=lambda o: bearing(n, o.split('_')[1]))
            links += [('%s_%s' % e, o) for o in outs]
        link_edges[n] = links
        ph = SMALL_GRID_PHASES[len(ins)]
        assert len(ph[0]) == len(links)
        phases.append(ph)
    kmax = max(len(v) for v in link_edges.values())
    demand = small_grid_demand(num_extra_car_per_hour, env_kw.get('episode_length_sec', 3600))
    route_paths = []
    for path, *_ in demand:
        if path not in route_paths:
            route_paths.append(path)
    NR = len(route_paths)
    mv_next = np.full((NL, NR), -2, np.int32)
    mv_link = np.full((NL, NR), -1, np.int32)
    lane_node = np.full(NL, -1, np.int32)
    for e in ['%s_%s' % x for x in edge_list]:
        if e.split('_')[1] in aidx:
            lane_node[piece_last[e]] = aidx[e.split('_')[1]]
    route_entry = np.zeros(NR, np.int32)
    for r, path in enumerate(route_paths):
        route_entry[r] = piece_first[path[0]]
        for p, e in enumerate(path):
            for l in range(piece_first[e], piece_last[e]):
                mv_next[l, r] = l + 1
            last = piece_last[e]
            if p + 1 == len(path):
                mv_next[last, r] = -1
            else:
                mv_next[last, r] = piece_first[path[p + 1]]
                n = e.split('_')[1]
                if n in aidx:
                    mv_link[last, r] = link_edges[n].index((e, path[p + 1]))
    lane_up = np.full((NL, MAX_UP), -1, np.int32)
    for l2 in range(NL):
        ups = sorted({l for l in range(NL) if (mv_next[l] == l2).any()})
        assert len(ups) <= MAX_UP
        lane_up[l2, :len(ups)] = ups
    A = len(node_names)
    link_lane = np.full((A, kmax), -1, np.int32)
    lanes_per_agent = []
    for a, n in enumerate(node_names):
        seq = [piece_last[e] for e, _ in link_edges[n]]
        link_lane[a, :len(seq)] = seq
        ded = []
        for l in seq:
            if l not in ded:
                ded.append(l)
        lanes_per_agent.append(ded)
    lmax = max(len(x) for x in lanes_per_agent)
    agent_lanes = np.full((A, lmax), -1, np.int32)
    for a, ded in enumerate(lanes_per_agent):
        agent_lanes[a, :len(ded)] = ded
    nlane = [len(x) for x in lanes_per_agent]
    neighbors = [[aidx[j] for j in SMALL_GRID_NEIGHBOR_MAP[n] if j in aidx] for n in node_names]
    n_a_ls = [len(p) for p in phases]
    a_max = max(n_a_ls)
    n_s, n_w, n_f = _state_dims(agent, nlane, n_a_ls, neighbors, True)
    obs_kind, obs_src, lens = _obs_table(agent, agent_lanes, nlane, n_a_ls, neighbors, True, a_max)
    green, yellow = _signal_tables(phases, kmax)
    rid = {p: i for i, p in enumerate(route_paths)}
    flows = np.array([[tb, te, vph, rid[path]] for path, tb, te, vph in demand], np.int32)
    stream_kw = {}
    if random_turns:
        # every vehicle of a source flow draws its path from the turn ratios, like SUMO's jtrrouter does per episode seed
        # (small_grid/data/build_file.py:223-335): one stream per source edge, one choice table per 600-s interval (the
        # ratios at npc vary with time, :283-293); the explicit-route "mf_" flows are one fixed-route stream per path
        src_elems, mf_elems = small_grid_demand(num_extra_car_per_hour, env_kw.get('episode_length_sec', 3600), distributions=True)
        srcs = []
        for src, *_ in src_elems:
            if src not in srcs:
                srcs.append(src)
        mf_paths = []
        for path, *_ in mf_elems:
            if path not in mf_paths:
                mf_paths.append(path)
        NI = max(i0 for _, i0, *_ in src_elems) + 1
        KC = max(len(pp) for *_, pp in src_elems)
        NS = len(srcs) + len(mf_paths)
        choice = np.full((NS, NI, KC, 2), -1, np.int32)
        choice[..., 1] = 0
        fl = []
        for src, i0, tb, te, vph, pp in src_elems:
            s_ = srcs.index(src)
            cum = np.cumsum([pr for _, pr in pp])
            for k, (path, _) in enumerate(pp):
                choice[s_, i0, k] = (rid[path], 65536 if k == len(pp) - 1 else int(round(65536 * cum[k] / cum[-1])))
            fl.append([tb, te, vph, s_])
        for i0 in range(NI):                      # intervals without a flow element keep the first table (never used)
            for s_ in range(len(srcs)):
                if choice[s_, i0, 0, 0] < 0:
                    choice[s_, i0] = choice[s_, 0]
        for path, tb, te, vph in mf_elems:
            s_ = len(srcs) + mf_paths.index(path)
            choice[s_, :, 0] = (rid[path], 65536)
            fl.append([tb, te, vph, s_])
        flows = np.array(fl, np.int32)
        stream_kw = dict(stream_entry_lane=np.array([piece_first[e] for e in srcs] + [piece_first[p[0]] for p in mf_paths], np.int32),
                         stream_mode=np.array([1] * len(srcs) + [0] * len(mf_paths), np.int32), stream_choice=choice,
                         choice_interval_sec=600)
    lane_det = np.where(lane_node >= 0, lane_len - DET_LEN, 0).astype(np.float32)
    scn = Scenario(
        name='small_grid', agent=agent, node_names=node_names, n_agent=A,
        lane_names=lane_names, lane_len=lane_len, lane_vmax=np.full(NL, 20.0, np.float32), lane_node=lane_node,
        lane_det_start=lane_det, lane_up=lane_up,
        n_route=NR, mv_next=mv_next, mv_link=mv_link, mv_yield=np.full((NL, NR), -1, np.int32),
        mv_prio=np.zeros((NL, NR), np.int32), mv_zip=np.zeros((NL, NR), np.int32), route_entry_lane=route_entry,
        route_names=[(p[0], p[-1]) for p in route_paths],
        agent_lanes=agent_lanes, agent_nlane=np.array(nlane, np.int32),
        agent_nlink=np.array([len(link_edges[n]) for n in node_names], np.int32), agent_nphase=np.array(n_a_ls, np.int32),
        link_lane=link_lane, phases=phases, green_tab=green, yellow_tab=yellow,
        neighbors=neighbors, n_s_ls=n_s, n_w_ls=n_w, n_f_ls=n_f, n_a_ls=n_a_ls,
        obs_kind=obs_kind, obs_src=obs_src, flows=flows, obs_len=lens,
        extra={'num_extra_car_per_hour': num_extra_car_per_hour, 'routes': route_paths, 'demand': demand,
               'state_phase_map': SMALL_GRID_STATE_PHASE_MAP}, **stream_kw, **env_kw)
    return sort_lanes_by_load(scn) if sort_lanes else scn



def lane_load(scn: Scenario) -> np.ndarray:
    """Vehicles per episode routed over every lane (static proxy for queue length)."""
    load = np.zeros(scn.n_lane)
    for tb, te, vph, s_ in scn.flows:
        veh = (te - tb) * vph / 3600.0
        if scn.stream_entry_lane is None:
            ways = [(int(scn.route_entry_lane[s_]), int(s_), 1.0)]
        else:                                   # a stream's vehicles spread over its routes (cumulative weights of 65536)
            NI = scn.stream_choice.shape[1]
            ways = []
            for i in range(NI):
                prev = 0
                for r, c in scn.stream_choice[s_, i]:
                    if r < 0:
                        break
                    ways.append((int(scn.stream_entry_lane[s_]), int(r), max(int(c) - prev, 1) / 65536.0 / NI))
                    prev = int(c)
        for l, r, w in ways:
            hops = 0
            while l >= 0 and hops <= 2 * scn.n_lane:
                load[l] += veh * w
                nx = int(scn.mv_next[l, r])
                if nx < -1 and scn.lane_sib is not None and scn.lane_sib[l] >= 0 and scn.mv_next[scn.lane_sib[l], r] >= -1:
                    nx = int(scn.lane_sib[l])           # rule 10: the vehicle moves over to the lane that serves its movement
                l = nx
                hops += 1
    return load


def permute_lanes(scn: Scenario, order) -> Scenario:
    """Renumber lanes: new lane i = old lane order[i].  Lane numbering is free (every table is
    index based); the HIP microsimulator maps thread -> lane, so putting the busiest lanes first
    packs long queues into the same wavefront and lets the other wavefronts retire early."""
    order = np.asarray(order)
    new_of = np.empty(len(order), np.int64)
    new_of[order] = np.arange(len(order))

    def remap(v):
        v = np.asarray(v)
        out = v.copy()
        m = v >= 0
        out[m] = new_of[v[m]]
        return out.astype(v.dtype)
    scn.lane_names = [scn.lane_names[i] for i in order]
    for k in ('lane_len', 'lane_vmax', 'lane_node', 'lane_det_start', 'lane_origin'):
        setattr(scn, k, getattr(scn, k)[order])
    if scn.lane_sib is not None:
        scn.lane_sib = remap(scn.lane_sib[order])
    up = remap(scn.lane_up[order])
    for i in range(len(up)):                                    # keep "ascending feeder index" order (rule 10: the sibling first)
        row = np.sort(up[i][up[i] >= 0])
        if scn.lane_sib is not None and scn.lane_sib[i] >= 0 and scn.lane_sib[i] in row:
            row = np.concatenate([[scn.lane_sib[i]], row[row != scn.lane_sib[i]]])
        up[i] = -1
        up[i, :len(row)] = row
    scn.lane_up = up
    scn.mv_next = remap(scn.mv_next[order])
    scn.mv_link = scn.mv_link[order]
    scn.mv_yield = remap(scn.mv_yield[order])
    scn.mv_prio = scn.mv_prio[order]
    scn.mv_zip = scn.mv_zip[order]
    scn.route_entry_lane = remap(scn.route_entry_lane)
    if scn.stream_entry_lane is not None:
        scn.stream_entry_lane = remap(scn.stream_entry_lane)
    scn.agent_lanes = remap(scn.agent_lanes)
    scn.link_lane = remap(scn.link_lane)
    src = scn.obs_src.copy()
    m = (scn.obs_kind >= 1) & (scn.obs_kind <= 3)
    src[m] = new_of[scn.obs_src[m]]
    scn.obs_src = src.astype(np.int32)
    return scn


def sort_lanes_by_load(scn: Scenario) -> Scenario:
    load = lane_load(scn)
    order = sorted(range(scn.n_lane), key=lambda i: (-load[i], i))
    return permute_lanes(scn, order)


def build_scenario(name: str, agent: str = 'ma2c', **kw) -> Scenario:
    if name == 'large_grid':
        return build_large_grid(agent, **kw)
    if name == 'real_net':
        return build_real_net(agent, **kw)
    if name == 'small_grid':
        return build_small_grid(agent, **kw)
    raise ValueError('unknown scenario %r (large_grid, real_net, small_grid)' % name)
