"""Is the step host-bound?  Compares the time to ENQUEUE a step (no sync) with the GPU time per step."""
import sys, time
sys.path.insert(0, '.')
import torch, bench
from desed_task_amd.arena import FusedAdam
from desed_task_amd.launcher import StepDriver
from desed_task_amd.nnet.CRNN import CRNN
from desed_task_amd.sed_trainer import SEDTask4
from desed_task_amd.utils.schedulers import ExponentialWarmup
dev = torch.device("cuda", 0)
config = bench.recipe_config()
student = CRNN(**config["net"]).to(dev)
opt = FusedAdam(student.parameters(), lr=1e-3, arena=student.arena)
sched = {"scheduler": ExponentialWarmup(opt, 1e-3, 5900), "interval": "step"}
class Enc: labels = list(range(10))
task = SEDTask4(config, Enc(), student, opt=opt, scheduler=sched).to(dev)
opt.arena = task.sed_student.arena
task.train()
driver = StepDriver(task, 1)
audio, labels = bench.synthetic_batch(dev, 1)
for i in range(5): driver.run_step((audio, labels.clone(), None, None), i)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N): driver.run_step((audio, labels.clone(), None, None), i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): driver.run_step((audio, labels.clone(), None, None), i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(25)
