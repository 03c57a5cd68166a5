"""tacotron_amd -- MI355X-native Tacotron acoustic-model hot path (host mirror of the reference's model object + drivers over
libtaco_hip.so).  See DESIGN.md / INTEGRATION.md."""
import os
import sys

# (see tacotron_amd/lib.py: must be in the environment before the HIP runtime initialises, i.e. before the first HIP API call --
#  which torch.cuda.is_available() / device_count() already are.  Recorded here, before the default is applied, so that lib.py can
#  tell a caller's own setting from this one, and an import that came first from one that may have come too late.)
QUEUES_SET_BY_USER = 'GPU_MAX_HW_QUEUES' in os.environ
TORCH_IMPORTED_FIRST = 'torch' in sys.modules
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
