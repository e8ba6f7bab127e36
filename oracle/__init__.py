"""TEST INFRASTRUCTURE ONLY: CPU oracles for the hot path (see oracle/*.c headers).

Nothing under openpcdet_amd/ imports this package; only tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py do.
"""
