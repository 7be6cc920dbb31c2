"""What would the step gain if a kernel family cost nothing? Runs bench.py's main() with one ops entry point replaced by a
no-op (TIMING ONLY: the results are wrong). Usage: whatif.py <name[,name...]> [bench.py arguments]
names: gram_bwd (the grouped Gram backward launches), gram_fwd (the grouped Gram forward), style (style_loss_grouped),
mse (masked content MSE)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
names = sys.argv[1].split(",")
sys.argv = [sys.argv[0]] + sys.argv[2:]
from stylemesh_amd.runtime import ops
table = {"gram_bwd": "gram_backward_grouped", "gram_fwd": "gram_masked_grouped", "style": "style_loss_grouped", "mse": "mse_masked"}
for n in names:
    if n != "none":
        setattr(ops, table[n], lambda *a, **k: None)
import bench
bench.main()
