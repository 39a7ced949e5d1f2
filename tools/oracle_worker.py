#!/usr/bin/env python3
"""One chain of a whole-host CPU baseline (bench.py `cpu_all_cores`, SURVEY.md §8d's optional third figure): loads the pickled job, calls the
oracle function again and again for the given number of seconds and prints `calls seconds`.  A process per chain, not a thread: threads of one
process were seen to share cores on the virtualised hosts of the pool.  Never touches the GPU.

  python tools/oracle_worker.py job.pkl chain_index"""
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

job = pickle.load(open(sys.argv[1], "rb"))
c = int(sys.argv[2])
fn = getattr(O, job["fn"])
args = list(job["args"])
kw = dict(job["kwargs"])
if job.get("chunks_arg") is not None:                 # every chain starts from its own configuration
    args[job["chunks_arg"]] = job["chunks"][c % len(job["chunks"])]
kw["replica"] = c
O.lib()
fn(*args, **dict(kw, it0=0)) if job.get("warm") else None
n, t0 = 0, time.perf_counter()
while True:
    fn(*args, **dict(kw, it0=n * job["it0_stride"]))
    n += 1
    dt = time.perf_counter() - t0
    if dt >= job["seconds"] or n >= 256:
        break
print(n, dt)
