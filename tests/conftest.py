import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SFG_ENABLE_TEST_HOOKS", "1")      # the library's failure-path test hook (sfg_ctx_encoder_inject_unsafe_for_test) is refused without it


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
