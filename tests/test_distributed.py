"""world_size-2 gloo test of the multi-GPU layer: contiguous sharding of the accepted-read stream and the exact
integer reduction of the per-family accumulators.  The sharded result must equal the unsharded one, and the
aggregates must equal the reference's aggregate_hits() on the golden best hits (ints identical, 'cov' sums to
1e-12 relative)."""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")

WORKER = r'''
import json, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from microbecensus_amd import distributed as D
from microbecensus_amd import _native, microbe_census as mc
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = json.load(open(os.path.join(sys.argv[1], "tests", "golden", "unittest_metagenome.json")))
model = _native.load_model(); fams = model["families"]
names, seqs = _native.load_markers()
items = sorted(g["best_hits"].items(), key=lambda kv: int(kv[0]))
reads = np.array([int(k) for k, _ in items])
lo, hi = D.shard_bounds(g["sampled_reads"], rank, world)          # shard the READ stream, not the hit list
mine = [(k, v) for k, v in items if lo <= int(k) < hi]
best = np.zeros(len(mine), dtype=_native.BEST_DTYPE)
for i, (k, v) in enumerate(mine):
    best[i] = (int(k), fams.index(v[0]), int(v[1]), int(round(v[1] / v[2])), v[3])
acc = D.family_accumulators(best, len(fams))
acc = D.all_reduce_accumulators(*acc)
agg = D.aggregate_from_accumulators(*acc, fams, mc.find_opt_pars(None, 100))
if rank == 0:
    json.dump({"agg": agg, "n": int(acc[0].sum())}, open(sys.argv[2], "w"))
dist.destroy_process_group()
'''


def test_two_rank_reduce_matches_reference_aggregates(tmp_path):
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29517", str(worker), REPO, str(out)], env=env, timeout=600)
    res = json.load(open(out))
    g = json.load(open(os.path.join(GOLD, "unittest_metagenome.json")))
    assert res["n"] == len(g["best_hits"])
    assert set(res["agg"]) == set(g["agg_hits"])
    for fam, want in g["agg_hits"].items():
        got = res["agg"][fam]
        if float(want).is_integer():
            assert got == want, fam
        else:
            assert abs(got - want) <= 1e-12 * abs(want), fam


def test_shard_bounds_cover_everything():
    from microbecensus_amd.distributed import shard_bounds
    for n in (0, 1, 7, 70623, 10**8):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
