"""Parity oracle package -- TEST INFRASTRUCTURE.  Import only from tests/, __graft_entry__.smoke()
and the cpu_baseline leg of bench.py; never from spliser_amd/."""
