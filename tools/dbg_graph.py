import sys, traceback, torch
sys.path.insert(0,''+__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))+''); sys.path.insert(0,''+__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))+'/dcs-net_amd')
from dcsnet.config import config, hparams
from dcsnet.c_network import C_NETWORK
from dcsnet.dp import TrainStep
from dcsnet import functional
from oracle.seeded_state import fill_state, seeded_input
dev=torch.device('cuda:0')
net = fill_state(C_NETWORK(config, hparams, 0), 2).to(dev).train()
clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
ts = TrainStep(net, use_graph=True, graph_warmup=2)
for i in range(2): ts(batch)
torch.cuda.synchronize()
try:
    ts._capture(batch)
    print('capture ok')
except Exception:
    traceback.print_exc()
