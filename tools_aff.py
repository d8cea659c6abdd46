import os
print("before", len(os.sched_getaffinity(0)))
import torch
print("after import", len(os.sched_getaffinity(0)), torch.get_num_threads())
torch.cuda.set_device(0); torch.cuda.synchronize()
print("after cuda init", len(os.sched_getaffinity(0)))
import threading
print("threads", threading.active_count(), len(os.listdir('/proc/self/task')))
