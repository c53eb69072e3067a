"""tools/v4_step_probe.py -- dev probe: the training step of the reference's network_v4 through the library's own layers
(do_trec_qa_clean.py:452-470 read as data): Embed x 2 (one shared 50-d table, bias) -> SimCross dist_mode 2, M = 4, bias
-> backward -> Embed backward x 2.  Batch 50, 40 words, vocabulary 20,000, sentences zero-padded to 40 words.
Run under rocprofv3 --kernel-trace --stats for the per-kernel breakdown (profiles/r03_v4_step_kernel_stats.txt)."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mms_answer_selection_amd import capi

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device="cuda").manual_seed(1701)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.4
Bt, Wd, Dv, Mv, Kv = 50, 40, 50, 4, 20000
tab, ebias = rnd(Kv, Dv), torch.zeros(Dv, device="cuda")
iq = torch.randint(0, Kv, (Bt, Wd), device="cuda", generator=g).float()
ia = torch.randint(0, Kv, (Bt, Wd), device="cuda", generator=g).float()
iq[:, 30:] = Kv - 1
ia[:, 34:] = Kv - 1
qe, ae = torch.empty(Bt, Wd, Dv, device="cuda"), torch.empty(Bt, Wd, Dv, device="cuda")
Wv = torch.rand(Mv, Dv, Dv, device="cuda", generator=g) * 0.16 - 0.08
bv = torch.zeros(Mv, Wd, Wd, device="cuda")
tv = torch.empty(Bt, Mv, Wd, Wd, device="cuda")
dtv = torch.randn(Bt, Mv, Wd, Wd, device="cuda", generator=g)
dqe, dae = torch.empty_like(qe), torch.empty_like(ae)
dWv, dbv = torch.empty_like(Wv), torch.zeros_like(bv)
dtab, debias = torch.zeros_like(tab), torch.zeros_like(ebias)
ws = capi.Workspace()
fused = os.environ.get("MMS_V4_PAIR", "1") == "1"


def step():
    capi.embed_forward(iq, tab, qe.view(-1, Dv), bias=ebias)
    capi.embed_forward(ia, tab, ae.view(-1, Dv), bias=ebias)
    capi.simcross_forward(2, qe, ae, tv, W=Wv, bias=bv, ws=ws)
    capi.simcross_backward(2, qe, ae, tv, dtv, dqe, dae, W=Wv, bias_term=True, dW=dWv, dbias=dbv, ws=ws)
    if fused:
        capi.embed_backward_pair(ia, iq, dae.view(-1, Dv), dqe.view(-1, Dv), dtab, bias_diff=debias, ws=ws)
    else:
        capi.embed_backward(iq, dqe.view(-1, Dv), dtab, bias_diff=debias, ws=ws)
        capi.embed_backward(ia, dae.view(-1, Dv), dtab, bias_diff=debias, ws=ws)


for _ in range(5):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    step()
e1.record()
torch.cuda.synchronize()
print("network_v4 training step, eager: %.1f us per step over %d steps" % (e0.elapsed_time(e1) * 1e3 / iters, iters))
