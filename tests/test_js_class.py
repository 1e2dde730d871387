"""The drop-in JavaScript `Homography` class (homography.js_amd/js/Homography.mjs) under Node.
CPU: state machine replay of every golden script (path chosen, output window, points, matrices), own Delaunay,
error strings, CSS export, loud failure without a GPU.  GPU: full replay incl. RGBA + triangle-map hashes."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "homography.js_amd", "lib", "hgwarp.node")

HAVE_JS = NODE is not None and os.path.exists(ADDON)
# CPU-side JS tests skip without Node; the GPU ones FAIL instead (below): on a GPU box a missing node / addon must not make
# the JavaScript drop-in's parity tests silently vanish.
needs_js = pytest.mark.skipif(not HAVE_JS, reason="node or the N-API addon is missing")


def _require_js_on_gpu():
    assert NODE is not None, "node is not on PATH on this GPU box: the JavaScript drop-in (the product's host side) cannot be tested"
    assert os.path.exists(ADDON), f"{ADDON} is missing: run `make -C homography.js_amd` (needs /usr/include/node/node_api.h)"


def _node(script, *args, timeout=900, env=None):
    p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", script), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, **env) if env else None)
    line = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert line, f"no JSON from {script}: rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}"
    return p.returncode, json.loads(line[-1])


@needs_js
def test_state_machine_replay_matches_reference():
    """Every golden script -- incl. the call sequences over the reference's stale caches, warpBatch against
    the reference's loop, the CSS strings and every bare-string error -- replayed on the class with its device calls answered by the
    JavaScript oracle (tests/js/mock_addon.cjs): the class's state machine and what it asks the native layer for, pixel for pixel."""
    rc, res = _node("replay_golden.mjs", "--dry", timeout=900)
    assert res["failures"] == [] and rc == 0
    assert res["cases"] >= 144 and res["warps"] >= 340 and res["staleStateWarps"] >= 40 and res["expectedThrows"] >= 30 and res["cssStrings"] >= 20


@needs_js
@pytest.mark.skipif(not os.path.exists("/root/reference/Homography.js"), reason="the live reference exists only in the build container")
def test_random_call_sequences_against_the_live_reference():
    """Differential fuzz: random call sequences on the reference's own Homography.js and on the class (over the mock addon), op by op."""
    rc, res = _node("fuzz_ref_sequences.mjs", "250", "11", timeout=900)
    assert res["failures"] == [] and rc == 0
    assert res["warps"] >= 800 and res["stateCalls"] >= 50


@needs_js
@pytest.mark.skipif(not os.path.exists("/root/reference/Homography.js"), reason="the live reference exists only in the build container")
@pytest.mark.parametrize("solve", ["reference", "addon"])
def test_reference_class_patched_over_the_addon(solve):
    """INTEGRATION.md section B, executed: the REFERENCE's own class with its four private loops replaced by addon calls exactly as
    homography.js_amd/js/reference_patch.mjs prints them (reference-state forms included), over the mock addon.  Every golden script (4K /
    8K included) gives the bytes the reference recorded, and 250 random call sequences give, op by op, the exception, state, path and
    bytes of the UNPATCHED reference.  The fast calls and the state forms must both have been exercised."""
    args = ["250", "11"] if solve == "reference" else ["250", "5", "--skip-big", "--own-solve"]
    rc, res = _node("ref_patched_over_addon.mjs", *args, timeout=900)
    assert res["failures"] == [] and rc == 0
    g, q = res["golden"], res["sequences"]
    assert g["cases"] >= (144 if solve == "reference" else 130) and g["warps"] >= 310 and g["staleStateWarps"] >= 40 and g["expectedThrows"] >= 30 and g["stateCalls"] >= 30
    assert q["sequences"] == 250 and q["warps"] >= 800 and q["stateCalls"] >= 50 and q["fastCalls"] >= 500


@needs_js
def test_host_side_javascript():
    rc, res = _node("test_host.mjs")
    assert res["failures"] == [] and rc == 0


@needs_js
def test_frame_pool_never_shares_a_backing_store_under_the_v8_8_rule():
    """Two ArrayBuffers over one backing pointer abort Node >= 14: with early reuse switched off (what the addon does by itself
    there) a pooled buffer is recycled only after its previous ArrayBuffer's finalizer has run."""
    rc, res = _node("test_pool_v8.mjs", env={"HGWARP_POOL_NO_EARLY_REUSE": "1"})
    assert res["failures"] == [] and rc == 0


@needs_js
def test_png_codec_on_reference_fixture():
    rc, res = _node("test_png.mjs")
    assert res["failures"] == [] and rc == 0


@needs_js
def test_warp_without_gpu_throws_a_string():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    code = """
import { Homography } from './homography.js_amd/js/Homography.mjs';
const h = new Homography('affine');
h.setReferencePoints([[0,0],[0,1],[1,0]], [[0,0],[0,2],[2,0]]);
try { h.warp({data: new Uint8ClampedArray(16*16*4), width: 16, height: 16}); console.log(JSON.stringify({threw: false})); }
catch (e) { console.log(JSON.stringify({threw: true, type: typeof e, msg: String(e)})); }
"""
    p = subprocess.run([NODE, "--input-type=module", "-e", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["threw"] and res["type"] == "string" and "no CPU fallback" in res["msg"]


@pytest.mark.gpu
def test_full_replay_on_gpu_matches_reference_hashes():
    _require_js_on_gpu()
    rc, res = _node("replay_golden.mjs", timeout=1500)
    assert res["failures"] == [] and rc == 0
    assert res["mode"] == "gpu" and res["warps"] >= 340 and res["staleStateWarps"] >= 40


@pytest.mark.gpu
def test_batch_and_buffer_aliasing_on_gpu():
    _require_js_on_gpu()
    rc, res = _node("test_gpu_batch.mjs", timeout=600)
    assert res["failures"] == [] and rc == 0


@pytest.mark.gpu
def test_reference_known_answer_png_through_js_class_on_gpu():
    """test/nodeTest.js flow on the reference's own input PNG == the reference's own expected output PNG, byte for byte."""
    _require_js_on_gpu()
    rc, res = _node("test_png.mjs", "--gpu", timeout=600)
    assert res["failures"] == [] and rc == 0
