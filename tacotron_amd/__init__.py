"""tacotron_amd -- MI355X-native Tacotron acoustic-model hot path (host mirror of the reference's model object + drivers over
libtaco_hip.so).  See DESIGN.md / INTEGRATION.md."""
import os

# (see tacotron_amd/lib.py: must be in the environment before the HIP runtime initialises, i.e. before the first device call)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
