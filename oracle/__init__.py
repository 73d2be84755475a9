"""TEST INFRASTRUCTURE ONLY: CPU oracle for the motion-estimation path.

Importable only from tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py. The shipped engine (x265_amd) never imports this package.
"""
