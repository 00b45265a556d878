"""Prints the elements whose post-Adam weight differs from the fixture's although the gradient signs agree."""
import os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import conftest  # noqa
from conftest import Golden
from adaptive_stereo.adaptation import OnlineAdapter
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
from adaptive_stereo.utils import synthetic as syn
DEV = "cuda:0"
case = sys.argv[1] if len(sys.argv) > 1 else "plumbing_240x320_k3_b1"
gold = Golden(case); meta = gold.meta
fnet = FeatureExtractorNetwork(meta["k"]); snet = StereoNet(meta["k"], 1, meta["s"], maxdisp=meta["maxdisp"])
init = {"feature": syn.synthetic_state_dict(fnet.state_dict(), seed=123),
        "stereo": syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=meta["gain"])}
fnet.load_state_dict(init["feature"]); snet.load_state_dict(init["stereo"])
fnet, snet = fnet.to(DEV), snet.to(DEV)
left, right = (t.to(DEV) for t in syn.stereo_pair(meta["B"], meta["H"], meta["W"], seed=1))
ad = OnlineAdapter(fnet, snet, meta["H"], meta["W"], lr=meta["lr"])
ad.step(left, right); torch.cuda.synchronize()
names = ("stereo", "feature")
nets = {"stereo": snet, "feature": fnet}
for mi, name, p, off, n in ad.arena.entries:
  gkey, akey = "grad/%s.%s" % (names[mi], name), "after/%s.%s" % (names[mi], name)
  if not gold.has(gkey):
    continue
  gref, full = gold.expected(gkey)
  g = ad.arena.grads[off:off + n].view(p.shape).detach().cpu()
  g = (g if full else syn.subsample(g, 4096)).reshape(gref.shape)
  aref, afull = gold.expected(akey)
  a = p.detach().cpu(); a = (a if afull else syn.subsample(a, 4096)).reshape(aref.shape)
  w0 = init[names[mi]][name]; w0 = (w0 if afull else syn.subsample(w0, 4096)).reshape(aref.shape)
  bad = ((a - aref).abs() > 5e-6) & (torch.sign(g) == torch.sign(gref)) & (gref.abs() >= 1e-6 * max(1.0, meta["gain"]))
  if bool(bad.any()):
    idx = bad.reshape(-1).nonzero().reshape(-1)[:4]
    for i in idx:
      i = int(i)
      print(names[mi], name, "full" if full else "sub", "afull" if afull else "asub", "elem", i, "g_ref %.3e g_gpu %.3e w0 %.6f after_ref %.6f after_gpu %.6f" % (
          float(gref.reshape(-1)[i]), float(g.reshape(-1)[i]), float(w0.reshape(-1)[i]), float(aref.reshape(-1)[i]), float(a.reshape(-1)[i])))
