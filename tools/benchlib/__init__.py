"""tools/benchlib -- the parts of bench.py (round 6: the 1 700-line file split by concern; bench.py keeps the contract, the command line,
run_workload and main, and re-exports these names).  Measurement infrastructure, not product code."""
