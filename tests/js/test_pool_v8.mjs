// Life cycle of the page-locked frame pool under the rule for V8 >= 8 (Node >= 14): a pooled buffer may back a new external
// ArrayBuffer only after the previous one's finalizer HAS run (two ArrayBuffers over one backing pointer abort newer Node
// versions, nodejs/node#32463).  Run with HGWARP_POOL_NO_EARLY_REUSE=1, which switches the Node 12 build of the addon to that
// rule; no GPU needed (_poolTestFrames: malloc memory through the allocation path of warp()).
import { createRequire } from 'module';
const require = createRequire(import.meta.url), hg = require('../../homography.js_amd/lib/hgwarp.node');
const fails = [];
const ok = (c, m) => { if (!c) fails.push(m); };
const gc = () => { const v8 = require('v8'), vm = require('vm'); v8.setFlagsFromString('--expose-gc'); vm.runInNewContext('gc')(); };
const B = 4 * 1024 * 1024;
(async () => {
    ok(hg.poolStats().earlyReuse === 0, 'HGWARP_POOL_NO_EARLY_REUSE=1 must switch early reuse off');
    // 1. a loop that never yields: dead frames are noticed (weak references) but NOT handed out again
    for (let it = 0; it < 24; it++) { const o = hg._poolTestFrames(1, B)[0]; o[0] = it; if (it % 8 === 7) { gc(); hg.poolCollected(); } }
    let st = hg.poolStats();
    ok(st.reapedByWeakRef > 0, `collections should have been noticed: ${JSON.stringify(st)}`);
    ok(st.reused === 0, `no buffer may be reused before its finalizer ran: ${JSON.stringify(st)}`);
    // 2. release(): detached at once, recycled only after the finalizer
    const r = hg._poolTestFrames(1, B)[0];
    ok(hg.release(r) === true && r.length === 0, 'release detaches');
    const again = hg._poolTestFrames(1, B)[0];
    ok(hg.poolStats().reused === 0, 'a released buffer is not recycled before its finalizer ran');
    again[0] = 1;
    // 3. one turn of the event loop: finalizers run, buffers come back
    gc();
    await new Promise((res) => setImmediate(res));
    await new Promise((res) => setImmediate(res));
    st = hg.poolStats();
    ok(st.finalized > 0, `finalizers should have run after an event-loop turn: ${JSON.stringify(st)}`);
    const live = hg._poolTestFrames(6, B);
    ok(hg.poolStats().reused > 0, `finalized buffers should be reused: ${JSON.stringify(hg.poolStats())}`);
    live.forEach((o, k) => o.fill(k + 1));
    ok(live.every((o, k) => o[0] === k + 1 && o[B - 1] === k + 1) && new Set(live.map((o) => o.buffer)).size === 6, 'live frames never share memory');
    console.log(JSON.stringify({ failures: fails }));
    process.exit(fails.length ? 1 : 0);
})();
