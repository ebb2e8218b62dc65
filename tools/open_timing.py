import sys, time
sys.path.insert(0, "/root/repo")
t0 = time.time()
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native
t1 = time.time()
model = _native.load_model()
t2 = time.time()
eng = _native.Engine(device=0)
t3 = time.time()
eng.set_run(150, model["pars"]["150"], model["families"])
t4 = time.time()
import numpy as np
reads = np.full((1000, 150), ord("A"), dtype=np.uint8)
eng.search(reads)
t5 = time.time()
print("import %.3f  load_model %.3f  Engine() %.3f  set_run %.3f  first search (pools) %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
