import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from tests.common import load_case, rms
from speech_inpainting_amd.engine import InpaintingEngine
c = load_case("base_4s"); z = c["z"]; m = c["meta"]
pos = torch.tensor(c["frame_pos"], dtype=torch.int32, device="cuda")
for enc, voc in (("fp32","fp32"),("fp32","bf16x3"),("fp32","bf16"),("bf16","fp32"),("bf16x3","bf16x3")):
    eng = InpaintingEngine(c["harch"], c["varch"], m["K"], "cuda:0", enc, voc).load_state(c["hsd"], c["gsd"], c["cb"])
    out = eng.predict_batch(c["wave"].cuda(), c["mel"].cuda(), pos, m["lm"])
    torch.cuda.synchronize()
    w = out["wave"].cpu()
    fe = rms(out["feats"].cpu(), z["feats"]) / rms(z["feats"])
    print(f"enc={enc:7s} voc={voc:7s} feats rel err {fe:.3e} labels agree {(out['labels'].cpu().numpy()==z['labels']).mean():.2f} wave rms err {rms(w, z['wave']):.3e} (signal rms {rms(z['wave']):.3f}) max abs err {float((w-torch.from_numpy(z['wave'])).abs().max()):.3e}")
    del eng
