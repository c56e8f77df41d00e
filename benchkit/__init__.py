"""Pieces of bench.py that are not the timed region: roofline models (roofline), the N > 1 launcher and the transport
negotiation (launch), the untimed extra workloads of the single-GPU line (extras).  bench.py keeps the argument parsing, the
timed steps, the CPU baseline (the one place outside tests/ that may load oracle/) and the JSON line."""
