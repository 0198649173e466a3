"""Debug aid: one wide-kernel test, a few times in one process."""
import sys, pytest
rc = 0
for i in range(4):
    r = pytest.main(["-q", "-x", "tests/test_gpu_wide.py::test_device_built_matrix_matches_oracle", "-p", "no:cacheprovider"])
    print("run", i, "rc", int(r))
    rc |= int(r)
sys.exit(rc)
