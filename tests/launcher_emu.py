"""TEST INFRASTRUCTURE: `python -m tests.launcher_emu <the launcher's arguments>` = desed_task_amd.launcher's command line on a CPU device
over gloo with the fiber-emulator build of the kernels bound (tests/emu) -- the plumbing check of tests/test_ddp_gloo.py.  The product
module never imports anything from tests/ (tests/test_abi.py::test_product_never_imports_oracle_or_emulator)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.emu_support import bind_emulator  # noqa: E402

bind_emulator()
from desed_task_amd import launcher  # noqa: E402

if __name__ == "__main__":
    sys.exit(launcher.main(cpu_test_device=True, entry_module="tests.launcher_emu"))
