"""wave-u-net_amd: MI355X-native Wave-U-Net hot path (forward + backward + Adam), see DESIGN.md.

The directory name carries a hyphen (it is the project's name); import it as
`wave_u_net_amd` (the alias package at the repo root re-exports this one)."""
from .config import BASE_MODEL_CONFIG, NAMED_CONFIGS, get_config, finalize      # noqa: F401
from .separator import UnetAudioSeparator                                         # noqa: F401
from . import _lib                                                                # noqa: F401
