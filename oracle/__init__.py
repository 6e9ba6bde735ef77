"""CPU oracle of the retrieval hot path: TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it; the product path under ccrec_amd/ never does)."""
