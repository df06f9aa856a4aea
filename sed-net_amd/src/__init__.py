"""Drop-in mirror of the reference's `src` package for the inference hot path.

Put `sed-net_amd/` on sys.path instead of the reference root and the call sites in
generate_predictions_aug.py (`from src.SEDNet import SEDNet`, `from src.mean_shift import MeanShift`, ...)
resolve to these MI355X-native implementations with the same names, signatures and return conventions.
"""
