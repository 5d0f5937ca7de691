from hawkeye_amd.types import Balltag, Ballarc  # noqa: F401
