import sys, torch, numpy as np
sys.path.insert(0, '.')
from paroquant_amd.decoder import ParoDecoderLM
from paroquant_amd import ops
dev = torch.device("cuda:0")
lm = ParoDecoderLM.random("qwen3-4b", dev, n_layers=4, max_positions=264)
print("fuse", lm.fuse_qkv_attn)
ids = torch.randint(0, 1000, (128,), device=dev)
lm.generate(ids, 40, use_graph=True)     # warm, graph; positions up to 168
torch.cuda.synchronize()
# the workspace holds the LAST layer-launch's stamps (every layer overwrites): good enough
w = lm.attn_ws.view(torch.uint8)[2048:].view(torch.int64).cpu().numpy()
att = w[:40 * 4].reshape(40, 4)          # per attention WG: poll start, polled, body start, body end
act = att[att[:, 2] > 0]
prod = w[4096:4096 + 2 * 1024].reshape(2, 512, 2)
pr = prod[prod[:, :, 1] > 0]
t0 = min(pr[:, 0].min(), act[:, 2].min())
print("producers: n", len(pr), "start min/max %.2f %.2f  end min/med/max %.2f %.2f %.2f us" % tuple(x / 100.0 for x in (pr[:, 0].min() - t0, pr[:, 0].max() - t0, pr[:, 1].min() - t0, np.median(pr[:, 1]) - t0, pr[:, 1].max() - t0)))
for r in act:
    print("attn WG: body start %.2f  poll start %.2f  polled %.2f  end %.2f" % tuple((x - t0) / 100.0 for x in (r[2], r[0], r[1], r[3])))
