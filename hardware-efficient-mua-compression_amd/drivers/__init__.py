"""GPU ports of the reference's three compression-evaluation scripts (callable, no Windows
paths): get_BR_no_sort, get_BR_with_approx_sort, test_chosen_system."""
from . import get_BR_no_sort, get_BR_with_approx_sort, test_chosen_system  # noqa: F401
