import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_runtest_teardown(item, nextitem):
    """ZKT_TEST_MEMINFO=<file>: append the device's free memory after every GPU test (leak hunting on the GPU box)."""
    path = os.environ.get("ZKT_TEST_MEMINFO")
    if not path or item.get_closest_marker("gpu") is None:
        return
    try:
        import torch
        free, total = torch.cuda.mem_get_info()
        with open(path, "a") as f:
            f.write(f"{free / 2**30:9.3f} GiB free of {total / 2**30:.1f}  torch reserved {torch.cuda.memory_reserved() / 2**30:.3f}  {item.nodeid[:110]}\n")
    except Exception as e:          # diagnostics only
        with open(path, "a") as f:
            f.write(f"meminfo failed: {e}\n")
