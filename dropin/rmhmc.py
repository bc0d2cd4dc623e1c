"""`from rmhmc import RMHMC` — put this directory in front of the reference's code/
directory on sys.path and an unchanged code/main.py imports the MI355X sampler
(main.py:12); swap main.py:52/53 to exercise it.  See INTEGRATION.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riemannhamiltonianmontecarlo_amd.rmhmc import RMHMC  # noqa: E402,F401
