"""Hook / log pipeline and time-overlap analysis of the reference's scripts, fed by device-side snapshots.

Mirrors scripts/scripts.jl: the hook of ``gen_hook`` (:51-69) keeps a copy of the configuration and prints one
``mctime acc E clocktime`` line per sample; ``to_mat`` (:13-21) packs the copies into a BitMatrix; ``parseovs``
(:368-405) turns them into <q^2>(t) over logarithmic time windows (``LogRange`` :336-358, ``get_ts_range`` :362-366).
Here the copies are snapshots in HBM (``Engine.snapshot_store``) and ``pm1dot`` (:283-295) runs on the device
(``Engine.overlaps``); only the Float64 statistics are finished on the host, in the reference's operation order.
"""
import time

import numpy as np

LRINCR = 1.5          # scripts/scripts.jl:347


class SnapshotLog:
    """``hook, cleanup, Cv = gen_hook(alg, seed)`` for a batch of replicas.

    ``hook(it, X, C, accepted, E)`` appends one line per replica to ``<prefix>_r<replica>.txt`` in the reference's
    format (header ``#mctime acc E clocktime``) and stores the live configuration in the next device snapshot slot.
    It returns ``t < t_limit`` like the reference's hook (:63-66).
    """

    def __init__(self, engine, nslots, prefix=None, t_limit=float("inf"), replicas=None, energy_label="E"):
        self.eng = engine
        self.nslots = int(nslots)
        engine.snapshot_reserve(self.nslots)
        self.t_limit = float(t_limit)
        self.t0 = time.time()
        self.mctimes, self.clock = [], []
        self.count = 0
        self.replicas = list(range(engine.R)) if replicas is None else list(replicas)
        self.files = {}
        if prefix is not None:
            for r in self.replicas:
                f = open("%s_r%d.txt" % (prefix, r), "w")
                f.write("#mctime acc %s clocktime\n" % energy_label)          # "QE" in test_QIsing (scripts.jl:802)
                self.files[r] = f

    def __call__(self, it, X, C, accepted, E):
        t = time.time() - self.t0
        if self.count >= self.nslots:
            raise RuntimeError("SnapshotLog: all %d snapshot slots are in use" % self.nslots)
        self.eng.snapshot_store(self.count)
        self.count += 1
        self.mctimes.append(it)                  # an iteration number; wtmMC hands its hook the sample's global time (src/RRRMC.jl:404)
        self.clock.append(t)
        for r, f in self.files.items():
            f.write("%s %d %s %s\n" % (_julia_num(it), int(accepted[r]), _julia_num(E[r]), repr(float(t))))
        return t < self.t_limit

    def close(self):
        for f in self.files.values():
            f.close()
        self.files = {}

    def to_mat(self, replica):
        """The reference's ``to_mat(Cv)`` for one replica: (bits[N, samples] as 0/1, chunks of the Julia BitMatrix)."""
        N = self.eng.X.N
        cols = np.zeros((N, self.count), np.uint8)
        for k in range(self.count):
            cols[:, k] = self.eng.snapshot_get(k).bits()[replica]
        return cols, bitmatrix_chunks(cols)


def _julia_num(x):
    """Julia prints Int energies as integers and Float64 ones with the shortest round-trip repr, as Python does."""
    return str(int(x)) if isinstance(x, (int, np.integer)) else repr(float(x))


def bitmatrix_chunks(cols):
    """Chunks of a Julia ``BitMatrix(N, samples)``: bit (i + N*j) of the flat, column-major bit string (no per-column padding)."""
    flat = np.asarray(cols, np.uint8).T.reshape(-1)          # column-major order of the [N, samples] matrix
    pad = (-flat.size) % 64
    flat = np.concatenate([flat, np.zeros(pad, np.uint8)])
    return np.packbits(flat, bitorder="little").view("<u8").copy()


def parsets(fname):
    """Clock times of a log written by ``SnapshotLog`` / the reference's hook (scripts/scripts.jl:297-333)."""
    ts = []
    with open(fname) as f:
        first = f.readline()
        assert first.startswith("#")
        for line in f:
            assert not line.startswith("#")
            sl = line.split()
            assert len(sl) >= 4
            ts.append(float(sl[3]))
    return ts


def log_range(i0, i1, st0=1.0, incr=LRINCR):
    """``LogRange(i0, i1, st0, incr)`` (scripts/scripts.jl:336-358) for Float64 endpoints."""
    assert st0 > 0 and incr >= 1
    i, st = i0, st0
    while not i > i1:
        yield i
        i, st = i + st, st * incr


def get_ts_range(tsm, t0):
    """0-based (i, j): first sample with t >= t0, first with t >= 2 t0 (None when absent); scripts/scripts.jl:362-366."""
    i = next((k for k, t in enumerate(tsm) if t >= t0), None)
    j = next((k for k, t in enumerate(tsm) if t >= 2 * t0), None)
    return i, j


def parseovs(engine, tsm, lr, nsamples=None):
    """<q^2> and its spread over the windows [t, 2t) of ``lr`` for every replica, from the device snapshots 0..nsamples-1.

    Returns (mq2s[windows, R], sq2s[windows, R]) — one column per replica where ``parseovs`` (scripts/scripts.jl:368-405)
    returns one vector for its single chain.  The pairs (i1, j1) are exactly the reference's loop ranges.
    """
    n = len(tsm) if nsamples is None else int(nsamples)
    N = engine.X.N
    mq2s, sq2s = [], []
    for t_st in lr:
        i, j = get_ts_range(tsm[:n], t_st)
        if i is None:
            break
        if j is None:
            j = n
        pa = [(i1, j1) for i1 in range(i, j - 1) for j1 in range(i + 1, j)]
        if not pa:
            mq2s.append(np.full(engine.R, np.nan))
            sq2s.append(np.full(engine.R, np.nan))
            continue
        mq2 = np.zeros(engine.R)
        mq4 = np.zeros(engine.R)
        for c0 in range(0, len(pa), 32768):
            blk = pa[c0:c0 + 32768]
            q = engine.overlaps([p[0] for p in blk], [p[1] for p in blk])
            for row in q:                                  # sequential accumulation, as the reference does
                q2 = (row / N) ** 2
                mq2 += q2
                mq4 += q2 ** 2
        mq2 /= len(pa)
        mq4 /= len(pa)
        mq2s.append(mq2)
        sq2s.append(np.sqrt(np.maximum(0.0, mq4 - mq2 ** 2)))
    R = engine.R
    return (np.array(mq2s).reshape(-1, R), np.array(sq2s).reshape(-1, R))
