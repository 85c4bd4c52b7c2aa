import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
import bench
bench.torch = torch
dev = torch.device("cuda")
for n in (256, 4096):
    roots = bench.beam_roots(ops, 4096, 0, dev)[:n].contiguous() if n < 4096 else bench.beam_roots(ops, 4096, 0, dev)
    if n < 4096:
        idx = torch.randperm(4096, generator=torch.Generator().manual_seed(1))[:n].to(dev)
        roots = bench.beam_roots(ops, 4096, 0, dev)[idx].contiguous()
    for _ in range(2):
        a, p, e = ops.beam_get_action(roots, 20, 30, seed=0x2048, step_index=11, want_expanded=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    a, p, e = ops.beam_get_action(roots, 20, 30, seed=0x2048, step_index=11, want_expanded=True)
    e1.record(); torch.cuda.synchronize()
    print("%s n=%d: kernel %.1f us; mean stage ticks/16 per decision %.0f (s_memtime 100 MHz ticks: x16 -> %.1f us)" % (
        os.path.basename(os.environ.get("G2048_LIB", "plain")), n, e0.elapsed_time(e1) * 1e3, e.float().mean().item(), e.float().mean().item() * 16 / 100.0))
