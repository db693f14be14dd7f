"""Round 4: diagnostic builds of the 128^2 kernel's GELU-from-global-table polynomial (pv_gemm.hip, PV_GELU_GLOBAL_MODE 0/2/3/4/5) for
scripts/dbg/gelu_glitch.py: libpeekvit_hip_gm<k>.so.  Run here (hipcc cross-compiles), the libraries travel with the snapshot."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from concurrent.futures import ThreadPoolExecutor
from peekvit_amd import _build
modes = [int(a) for a in sys.argv[1:]] or [0, 2, 3, 4, 5]
with ThreadPoolExecutor(3) as ex:
    for lib in ex.map(lambda k: _build.build_variant(f"gm{k}", [f"-DPV_GELU_GLOBAL_MODE={k}"]), modes):
        print(lib, flush=True)
