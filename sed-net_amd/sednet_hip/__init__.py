"""MI355X-native kernels for SED-Net's inference hot path (ctypes over libsedhip.so)."""
