"""CPU oracle (test infrastructure only -- see oracle/hhe_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / reported CPU baseline.
"""
from .oracle import *  # noqa: F401,F403
