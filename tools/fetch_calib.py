#!/usr/bin/env python
"""Fold the two rocprofv3 --pmc passes over tools/fetch_calib (FETCH_SIZE, WRITE_SIZE; rocpd databases) into
profiles/rNN_fetch_calibration.json: factor = known bytes / reported bytes per access pattern.  bench.py multiplies the
committed FETCH_SIZE / WRITE_SIZE of a kernel by the factor of ITS access width before comparing with algorithmic bytes.

    python tools/fetch_calib.py fetch.db write.db profiles/r06_fetch_calibration.json
"""
import json
import sqlite3
import sys

KNOWN = 1 << 30


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, val in c.execute('select kernel_name, value from counters_collection where counter_name = ?', (counter,)):
        if 'calib_' not in name:
            continue
        k = name[name.index('calib_') + 6:].split('(')[0]
        tot, n = out.get(k, (0.0, 0))
        out[k] = (tot + float(val), n + 1)
    return {k: t / n for k, (t, n) in out.items()}


def main(fetch_db, write_db, out_path):
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    doc = {'what': 'tools/fetch_calib.hip: every kernel streams 1 GiB once; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; '
                   'reported_kb = average per launch; factor = known bytes / (reported_kb x 1024)',
           'known_bytes': KNOWN, 'fetch': {}, 'write': {}}
    for k, kb in sorted(f.items()):
        if k.startswith('read') and kb > 0:
            doc['fetch'][k] = {'reported_kb': kb, 'factor': KNOWN / (kb * 1024.0)}
    for k, kb in sorted(w.items()):
        if k.startswith('write') and kb > 0:
            doc['write'][k] = {'reported_kb': kb, 'factor': KNOWN / (kb * 1024.0)}
    json.dump(doc, open(out_path, 'w'), indent=1)
    print(json.dumps(doc))


if __name__ == '__main__':
    main(*sys.argv[1:4])
