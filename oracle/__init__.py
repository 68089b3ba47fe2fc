"""oracle/ -- TEST INFRASTRUCTURE, not product.

CPU restatement of the reference's pretrain hot path (restatement.py), the deterministic input/weight
recipe (recipe.py) and the generator of the golden fixtures (gen_golden.py, imports the reference;
build container only).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  madeleine_amd/ never does.
"""
