"""Development aid (GPU box): file -> AGS of one FASTQ.gz through run_pipeline_distributed with W ranks on this box's ONE GPU (gloo), the
.gz decoded on every rank (chunk slices) against rank 0 inflating alone and dealing (MC_DIST_GZ=0).  The ranks share the box's
CPU quota, so this shows what the chain costs, not what N hosts' worth of cores give.  python tools/dist_gz_probe.py [nreads] [W ...]"""
import json, os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
worlds = [int(a) for a in sys.argv[2:]] or [2, 4]
gen = synth.GenomeReads(device="cpu", seed=20261001)
td = tempfile.mkdtemp(prefix="mc_dgz_")
path = os.path.join(td, "reads.fq.gz")
bench.write_fastq(gen, n, 150, path, True)
worker = os.path.join(td, "w.py")
open(worker, "w").write(r'''
import contextlib, io, json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from microbecensus_amd import distributed as D
dist.init_process_group(backend="gloo")
w = []
for rep in range(3):
    dist.barrier()
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        est, args = D.run_pipeline_distributed({"seqfiles": [sys.argv[2]], "nreads": int(sys.argv[3]), "read_length": 150}, device=0)
    dist.barrier()
    w.append(time.time() - t)
if dist.get_rank() == 0:
    print(json.dumps({"wall": min(w[1:]), "est": est, "sampled": args["sampled_reads"], "dealt": D.run_pipeline_distributed.last_trace is not None}), flush=True)
dist.destroy_process_group()
''')
port = 29600
for W in worlds:
    for gz in ("1", "0"):
        port += 1
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MC_DIST_GZ=gz)
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(W), "--master-addr", "127.0.0.1", "--master-port", str(port),
                            worker, REPO, path, str(n)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
        if not lines:
            print("world %d MC_DIST_GZ=%s failed: %s" % (W, gz, p.stderr.decode()[-1500:]), flush=True); continue
        if os.environ.get("MC_DIST_TRACE"):
            print("\n".join([l for l in p.stderr.decode().splitlines() if l.startswith(("gz slice", "gz part"))][-2 * (n // 550000 + 2):]), flush=True)
        r = json.loads(lines[-1])
        print("world %d  %-34s %.3f s = %5.2f M reads/s  sampled %d  AGS %.3f" % (W, "every rank inflates its slices" if not r["dealt"] else "rank 0 inflates and deals", r["wall"], n / r["wall"] / 1e6, r["sampled"], r["est"]), flush=True)
