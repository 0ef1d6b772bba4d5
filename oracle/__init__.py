"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product path (codecad_amd) never does and fails loudly without its HIP
extension.  See oracle/sdf_oracle.c for the restatement and DESIGN.md for how it is pinned.
"""
from .oracle import *  # noqa: F401,F403
