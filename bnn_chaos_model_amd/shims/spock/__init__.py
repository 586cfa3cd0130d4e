"""Import shim for `from spock import FeatureRegressor, FeatureRegressorXGB` (figures/multiswag_5_planet.py:28-29).
The reference's `spock` package (figures/spock/__init__.py) also pulls in its N-body and XGBoost classifiers; only the
MultiSWAG regressor is on the accelerated path, the others are import-compatible placeholders that say so."""
from bnn_chaos_model_amd.regression import FeatureRegressor, data_setup_kernel  # noqa: F401

__version__ = "bnn_chaos_model_amd"


def _out_of_scope(name):
    class _Placeholder(object):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"spock.{name} is outside the MI355X MultiSWAG path (SURVEY.md section 2); "
                                      "use the reference implementation for it")
    _Placeholder.__name__ = _Placeholder.__qualname__ = name
    return _Placeholder


FeatureRegressorXGB = _out_of_scope("FeatureRegressorXGB")
FeatureClassifier = _out_of_scope("FeatureClassifier")
NbodyRegressor = _out_of_scope("NbodyRegressor")

__all__ = ["FeatureRegressor", "FeatureRegressorXGB", "FeatureClassifier", "NbodyRegressor"]
