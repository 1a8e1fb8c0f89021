"""CPU oracle for the GP-posterior + acquisition hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``robo_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may.  See ``oracle/gp_oracle.py`` for the parity status
of each half (acquisition half: pinned against the reference's own code;
george half: "parity unpinned", kernel definitions are this project's stated
contract).
"""
from .gp_oracle import *  # noqa: F401,F403
