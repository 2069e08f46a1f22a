"""revers-o embed + search hot path, MI355X-native."""
from .config import PEConfig, VARIANTS, DEFAULT_VARIANT, get_config, available_configs  # noqa: F401
