"""Drop-in alias: `import openmg` resolves to the MI355X implementation in `openmg_amd`.

Same public names as tsbertalan/openmg (mgSolve, mgCycle, defaults, smooth,
smoothToThreshold, coarseSolve, tools, operators, solvers) plus the alias `mg_cycle`."""
import sys

import openmg_amd
from openmg_amd import (clear_cache, coarseSolve, defaults, mg_cycle, mgCycle, mgSolve, operators,  # noqa: F401
                        smooth, smoothToThreshold, solvers, tools)

# `from openmg import operators`, `import openmg.tools` ... keep working
for _name in ("operators", "solvers", "tools"):
    sys.modules[__name__ + "." + _name] = getattr(openmg_amd, _name)
