"""`from hmc import HMC` for the reference's unchanged code/main.py (main.py:10,53): see INTEGRATION.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riemannhamiltonianmontecarlo_amd.hmc import HMC  # noqa: E402,F401
