"""Import-name shim: the reference's callers (worker.py, track.py, pickled
notes) refer to ``hawkeye.tracker`` and ``hawkeye.types``; both resolve to the
MI355X implementation in ``hawkeye_amd``.  See INTEGRATION.md."""
