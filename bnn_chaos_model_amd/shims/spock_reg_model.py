"""Import shim: put this directory on sys.path (or PYTHONPATH) AHEAD of the reference checkout and the evaluation scripts'
`import spock_reg_model` (figures/main_figures.py, figures/spock/regression.py:15) resolves to the MI355X implementation.
Every public name of the reference module that the inference path uses is re-exported."""
from bnn_chaos_model_amd.spock_reg_model import *  # noqa: F401,F403
from bnn_chaos_model_amd.spock_reg_model import __all__  # noqa: F401
