"""Diagnostics: GraphedStepDriver (desed_task_amd/graph.py) vs the eager StepDriver -- step time and capture sanity.
usage: python tools/graph_probe.py [--no-mixup] [--steps N]"""
import sys, time
sys.path.insert(0, '.')
import torch, bench
from desed_task_amd.arena import FusedAdam
from desed_task_amd.graph import GraphedStepDriver
from desed_task_amd.nnet.CRNN import CRNN
from desed_task_amd.sed_trainer import SEDTask4
from desed_task_amd.utils.schedulers import ExponentialWarmup
dev = torch.device("cuda", 0)
config = bench.recipe_config()
if "--no-mixup" in sys.argv:
    config["training"]["mixup"] = None
N = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 30
student = CRNN(**config["net"]).to(dev)
opt = FusedAdam(student.parameters(), lr=1e-3, arena=student.arena)
sched = {"scheduler": ExponentialWarmup(opt, 1e-3, 5900), "interval": "step"}
class Enc: labels = list(range(10))
task = SEDTask4(config, Enc(), student, opt=opt, scheduler=sched).to(dev)
opt.arena = task.sed_student.arena
task.train()
driver = GraphedStepDriver(task, 1, warmup=3)
audio, labels = bench.synthetic_batch(dev, 1)
for i in range(3):
    driver.run_step((audio, labels.clone(), None, None), i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): driver.eager.run_step((audio, labels.clone(), None, None), i)
torch.cuda.synchronize()
print("eager: %.3f ms/step" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
driver.run_step((audio, labels.clone(), None, None), 0)
torch.cuda.synchronize()
print("captured + first replay ok", flush=True)
t0 = time.perf_counter()
for i in range(N): driver.run_step((audio, labels.clone(), None, None), i)
torch.cuda.synchronize()
print("graph: %.3f ms/step, loss %.4f, step_num %d" % ((time.perf_counter() - t0) / N * 1e3, float(driver.loss.detach()),
                                                       sched["scheduler"].step_num), flush=True)
