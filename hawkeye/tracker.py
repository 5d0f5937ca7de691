from hawkeye_amd.tracker import (VideoScanner, ScannerException,  # noqa: F401
                                 play_video)
