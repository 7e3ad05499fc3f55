"""CPU oracle for the MSMDFusion sparse-voxel hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under msmdfusion_amd/ does (tests/test_boundary.py
enforces it).  See oracle/msmd_oracle.c for what is restated and how it is
pinned against the reference.
"""
