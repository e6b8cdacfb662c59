import os, sys, torch, warnings
sys.path.insert(0, "/root/repo"); warnings.filterwarnings("ignore")
import qat_vit_amd
torch.manual_seed(0)
t = qat_vit_amd.create_teacher("vit", num_classes=10).cuda().eval()
x = torch.randn(8, 3, 224, 224).cuda()
with torch.no_grad():
    out = t(x)
    ref = t.double().cpu().float() if False else None
t64 = qat_vit_amd.create_teacher("vit", num_classes=10)
t64.load_state_dict(t.state_dict()); t64 = t64.double().eval()
with torch.no_grad():
    r = t64(x.cpu().double())
print("passes", os.environ.get("QATVIT_TEACHER_PASSES", "3"), "rel L2 vs fp64:", ((out.cpu().double() - r).norm() / r.norm()).item())
