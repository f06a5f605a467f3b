import sys, torch
sys.path.insert(0, "/root/repo")
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import _lib
from tests.golden_util import Golden
from tests.hip_harness import make_job
g = Golden(sys.argv[1])
k = int(sys.argv[2])
whole = make_job(g, 0)
nm.JobSet([whole]).grads(0, export=False)
job = make_job(g, 0)
js = nm.JobSet([job])
js.grads(0, rowsplit=k)
js.check_split_errors(block=True)
torch.cuda.synchronize()
d = (job.grads - whole.grads).abs().max().item()
print("max |grad diff| vs whole-batch:", d, "max grad", whole.grads.abs().max().item())
