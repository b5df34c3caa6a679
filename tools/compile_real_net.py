#!/usr/bin/env python
"""Compile the Monaco ("real_net") scenario inputs into the small table file the package ships
(build container only: reads /root/reference).

Sources, all from the reference tree:
  real_net/data/in/most.net.xml ....... edges, lanes (length, speed, vehicle classes), lane-to-lane
                                        connections with tl / linkIndex
  envs/real_net_env.py:20-68 .......... NODES (phase key + directed neighbour list), PHASES
  real_net/data/build_file.py:15-105 .. the 16 (from, to, via) flows and their activation pattern

Output: deeprl_signal_control_amd/data/real_net.json -- only what the routes and the 28 signalised
nodes touch (no geometry, no internal lanes).  scenario.build_real_net() turns it into the dense
tables; tests/test_real_net.py pins them against the reference's own RealNetEnv.

    python tools/compile_real_net.py
"""
import heapq
import json
import os
import re
import sys
import xml.etree.ElementTree as ET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'deeprl_signal_control_amd', 'data', 'real_net.json')


def passenger_ok(lane):
    a, d = lane.get('allow'), lane.get('disallow')
    if a is not None:
        return 'passenger' in a.split()
    if d is not None:
        return 'passenger' not in d.split()
    return True


def main():
    root = ET.parse(os.path.join(REF, 'real_net/data/in/most.net.xml')).getroot()
    edges = {}
    for e in root:
        if e.tag == 'edge' and e.get('function') is None:
            lanes = [dict(index=int(l.get('index')), length=float(l.get('length')), speed=float(l.get('speed')),
                          ok=passenger_ok(l)) for l in e if l.tag == 'lane']
            edges[e.get('id')] = dict(frm=e.get('from'), to=e.get('to'), lanes=lanes)
    cons, all_tl = [], []
    for c in root:
        if c.tag == 'connection' and not c.get('from').startswith(':'):
            f, t, fl, tl_ = c.get('from'), c.get('to'), int(c.get('fromLane')), int(c.get('toLane'))
            if c.get('tl') and f in edges:
                all_tl.append((c.get('tl'), int(c.get('linkIndex')), '%s_%d' % (f, fl), f))
            if f in edges and t in edges and edges[f]['lanes'][fl]['ok'] and edges[t]['lanes'][tl_]['ok']:
                cons.append(dict(frm=f, to=t, fl=fl, tl=tl_, node=c.get('tl') or '',
                                 link=int(c.get('linkIndex')) if c.get('tl') else -1))
    # reference constants (imported from the reference module through the test stubs)
    from oracle import fake_traci
    from deeprl_signal_control_amd.scenario import build_large_grid
    fake_traci.install(build_large_grid())
    from envs import real_net_env
    nodes = {k: dict(phase=v[0], neighbors=list(v[1])) for k, v in real_net_env.NODES.items()}
    phases = {k: list(v) for k, v in real_net_env.PHASES.items()}
    from real_net.data import build_file
    rou = build_file.output_flows(1)                       # flow_rate 1: structure only
    flows = [dict(frm=m.group(1), to=m.group(2), via=m.group(3).split(), begin=int(m.group(4)), end=int(m.group(5)))
             for m in re.finditer(r'from="(\S+)" to="(\S+)" via="([^"]*)" begin="(\d+)" end="(\d+)"', rou)]
    # edge-level routing (free-flow time, Dijkstra), through the via way-points in order
    succ = {}
    for c in cons:
        succ.setdefault(c['frm'], set()).add(c['to'])

    def cost(e):
        return edges[e]['lanes'][0]['length'] / edges[e]['lanes'][0]['speed']

    def shortest(a, b):
        dist, prev, pq = {a: 0.0}, {}, [(0.0, a)]
        while pq:
            d, u = heapq.heappop(pq)
            if u == b:
                break
            if d > dist.get(u, 1e18):
                continue
            for v in sorted(succ.get(u, ())):
                nd = d + cost(v)
                if nd < dist.get(v, 1e18):
                    dist[v], prev[v] = nd, u
                    heapq.heappush(pq, (nd, v))
        path = [b]
        while path[-1] != a:
            path.append(prev[path[-1]])
        return path[::-1]

    routes, route_id = [], {}
    for f in flows:
        key = (f['frm'], f['to'], tuple(f['via']))
        if key not in route_id:
            way = [f['frm']] + f['via'] + [f['to']]
            path = [way[0]]
            for a, b in zip(way[:-1], way[1:]):
                path += shortest(a, b)[1:]
            route_id[key] = len(routes)
            routes.append(path)
        f['route'] = route_id[key]
    # keep: every edge on a route + every edge with a signal-controlled connection of a NODES junction
    keep = set(e for p in routes for e in p)
    for node, _, _, f in all_tl:
        if node in nodes:
            keep.add(f)
    out = dict(
        edges={e: dict(to=edges[e]['to'], lanes=[[l['length'], l['speed'], int(l['ok'])] for l in edges[e]['lanes']])
               for e in sorted(keep)},
        connections=[[c['frm'], c['fl'], c['to'], c['tl'], c['node'], c['link']] for c in cons
                     if c['frm'] in keep and (c['to'] in keep)],
        tl_links={n: {} for n in nodes}, nodes=nodes, phases=phases, routes=routes,
        flows=[[f['route'], f['begin'], f['end']] for f in flows])
    for node, link, lane, _ in all_tl:                    # signal link -> incoming lane (getControlledLanes)
        if node in nodes:
            out['tl_links'][node][str(link)] = lane
    # foes (round 4, junction interiors): signal link k of a node -> the links whose paths cross or join its path inside the
    # junction, from the junction's own right-of-way matrix (<request index foes>); the request index of a connection is the
    # position of its internal `via` lane in the junction's intLanes list
    junc = {}
    for j in root:
        if j.tag == 'junction' and j.get('intLanes'):
            junc[j.get('id')] = dict(int_lanes=j.get('intLanes').split(),
                                     foes={int(r.get('index')): r.get('foes') for r in j if r.tag == 'request'})
    via_of = {}                                            # (tl node, link) -> (junction, request index)
    for c in root:
        if c.tag == 'connection' and c.get('tl') in nodes and c.get('via'):
            jid = c.get('via')[1:].rsplit('_', 2)[0]
            if jid in junc and c.get('via') in junc[jid]['int_lanes']:
                via_of[(c.get('tl'), int(c.get('linkIndex')))] = (jid, junc[jid]['int_lanes'].index(c.get('via')))
    out['foes'] = {n: {} for n in nodes}
    for (node, link), (jid, q) in sorted(via_of.items()):
        bits = junc[jid]['foes'].get(q, '')
        foe_req = {i for i, ch in enumerate(reversed(bits)) if ch == '1'}
        out['foes'][node][str(link)] = sorted(k2 for (n2, k2), (j2, q2) in via_of.items() if n2 == node and j2 == jid and q2 in foe_req and k2 != link)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('%d edges, %d connections, %d routes (%s edges), %d flow elements -> %s (%d bytes)'
          % (len(out['edges']), len(out['connections']), len(routes), [len(p) for p in routes], len(flows), OUT,
             os.path.getsize(OUT)))


if __name__ == '__main__':
    main()
