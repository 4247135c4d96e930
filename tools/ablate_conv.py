import ctypes as C, sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vstnet_amd import _lib
v = sys.argv[1]
_lib.LIB_PATH = os.path.join(_lib.PKG_DIR, f"libvstnet_abl{v}.so")
from models.RevResNet import RevResNet
from vstnet_amd.synth import synthetic_state_dict, synthetic_frames
net = RevResNet(); net.load_state_dict(synthetic_state_dict()); net = net.cuda().eval()
x = synthetic_frames(1, 1024, 1024).cuda()
L = _lib.lib()
for cin, cout in ((256, 64), (64, 256), (64, 64)):
    for _ in range(2): net(x)
    _lib.check(L.vst_profile_begin(_lib.kernel_id(cin, cout, 1), 4096), "b")
    for _ in range(5): net(x)
    tot, n = C.c_double(0), C.c_int(0)
    _lib.check(L.vst_profile_end(C.byref(tot), C.byref(n)), "e")
    print(f"ablate={v} conv<{cin},{cout}>: {tot.value / n.value * 1e3:.1f} us over {n.value} launches")
