"""cProfile of InferenceKeypointsModel.infer_images on the GPU box: python tools/api_profile.py"""
import cProfile, importlib, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pytorch-human-pose_amd")
net = pkg.HigherHRNet(17, 32)
net.load_state_dict({k: torch.from_numpy(v) for k, v in pkg.synth.synth_passthrough_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 17, 0, tag_gain=8.0).items()})
model = pkg.InferenceKeypointsModel(net, det_thr=0.05, tag_thr=0.5, use_flip=False, input_size=512, device="cuda:0")
images = pkg.synth.synth_passthrough_raw_u8(64, 128, 128, 10, 17, 0) * 4
model.infer_images(images[:64])
pr = cProfile.Profile(); pr.enable()
model.infer_images(images)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
