"""
Test double for mixemt_amd.em.EmPlan on CPU tensors: same surface
(alloc_props / alloc_state / read_state / em_iter / finalize), arithmetic from
the ORACLE's em_step.  It exists so the multi-process orchestration in
mixemt_amd.dist (sharding, all-reduce placement, stop decision) can be driven
under gloo without a GPU.  Lives in tests/: the product never imports it.
"""
import numpy
import torch

from oracle import em_oracle


class CpuPlan(object):
    def __init__(self, mat, wts):
        self.mat = numpy.ascontiguousarray(mat, dtype=numpy.float64)
        self.wts = numpy.ascontiguousarray(wts, dtype=numpy.float64)
        self.n_rows, self.n_haps = self.mat.shape
        self.calls = 0
        self.restart_steps = 0          # restarts handed to em_iter, summed over calls
        self.widest = 0                 # most restarts one em_iter call carried

    def alloc_props(self, host):
        return torch.from_numpy(numpy.array(host, dtype=numpy.float64))

    def alloc_state(self, n_runs):
        return torch.zeros((n_runs, 3), dtype=torch.float64)      # done, iters, l1

    def read_state(self, state):
        return [(int(s[0]), int(s[1]), float(s[2])) for s in state]

    def em_iter(self, props, ln_props, state, colsum):
        self.calls += 1
        self.restart_steps += props.shape[0]
        self.widest = max(self.widest, props.shape[0])
        for b in range(props.shape[0]):
            if state is not None and state[b, 0] != 0:
                continue
            lnp = ln_props[b].numpy()
            mix, _ = em_oracle.em_step(self.mat, self.wts, lnp, numpy.empty_like(self.mat))
            # unscaled sums T_h = sum_r w_r exp(posterior_rh) / p_h
            colsum[b] = torch.from_numpy((self.wts[:, None] * numpy.exp(mix - lnp[None, :])).sum(axis=0))

    def finalize(self, colsum, ln_cur, ln_new, props_cur, state, tol, max_iter):
        for b in range(props_cur.shape[0]):
            if state[b, 0] != 0:
                continue
            total = float((props_cur[b] * colsum[b]).sum())
            new = ln_cur[b] + torch.log(colsum[b]) - numpy.log(total)
            ln_new[b] = new
            l1 = float((torch.exp(new) - props_cur[b]).abs().sum())
            state[b, 1] += 1
            state[b, 2] = l1
            if l1 < tol:
                state[b, 0] = 1
            elif state[b, 1] >= max_iter:
                state[b, 0] = 2
            else:
                ln_cur[b] = new
                props_cur[b] = torch.exp(new)
