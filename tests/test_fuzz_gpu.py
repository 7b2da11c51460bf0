"""Short, fixed-seed runs of the randomised checks under tools/probe/ (their long runs are logged in profiles/r03_*_fuzz.txt): every
kernel family against float64 / the oracle on shapes and parameters nobody picked by hand -- contraction engine (layouts, modes,
epilogues, tiles, planes), fused attention (ragged bags, head dims, dropout), segmented pooling (online softmax under extreme score
spreads), GENConv on random graphs, LayerNorm-mean16 / gated pool / Adam / concordance index, the adversarial step against the oracle
(backbones, losses, visibility, discriminators), the slab pad, two ranks against one process, and the supervised baselines against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("script,args", [("gemm_fuzz.py", ("80", "101")), ("attn_fuzz.py", ("24", "102")), ("pool_fuzz.py", ("40", "103")),
                                         ("graph_fuzz.py", ("40", "104")), ("misc_fuzz.py", ("10", "105")), ("oracle_fuzz.py", ("8", "106")),
                                         ("pad_fuzz.py", ("1", "107")), ("dp_fuzz.py", ("2", "108")), ("baseline_fuzz.py", ("6", "109"))])
def test_randomised_probe(script, args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env["ADVMIL_GEMM_MODE"] = "exact"          # the probes' starting arithmetic (those that cover bf16x3 select it themselves)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", script), *args], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=850)
    assert r.returncode == 0 and "all ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
