"""mirage_amd: MI355X-native (gfx950) engine for the MIRAGE MultiMAE pretraining hot path.

Host modules mirror the reference's nn.Module surface (mirage/model.py, mirage/utils.py,
mirage/input_adapters.py, mirage/output_adapters.py, mirage/criterion.py); all device math runs in
hand-written HIP kernels behind the C ABI of libmirage_hip.so (include/mirage_hip.h).
"""
__version__ = "0.1.0"
