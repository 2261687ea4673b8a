from typing import Any

STEP_OUTPUT = Any
