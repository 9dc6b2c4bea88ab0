"""GPU box: LM iteration time of C4 through the sharded code path with a world_size-1 native RCCL communicator
(the collectives are issued, there is nobody to talk to): the overhead of the path itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from pytheiasfm_amd import ba, synth, distributed as tdist
dist.init_process_group("gloo", rank=0, world_size=1)
p = synth.ba_config("C4")
o = ba.default_options(); o.max_num_iterations = 8
o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = 0.0; o.use_inner_iterations = 0
for mode in ("plain", "rccl"):
    h = ba.BaHandle(p.copy(), o)
    comm = None
    if mode == "rccl":
        comm = tdist.NativeRccl(0, 1); comm.attach(h)
    h.reset(p); h.snapshot(); h.restore(); h.run(trace_capacity=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        h.restore(); s, _ = h.run(trace_capacity=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "%.3f ms / LM iteration" % (1e3 * dt / (4 * s.num_iterations)), "final cost %.9e" % s.final_cost, flush=True)
    h.close()
    if comm: comm.close()
