"""torch float64 autograd twin of the INTENDED `_train` step (/root/reference/src/neural_net.jl:75-101,
Momentum(2f-2) of train.jl:54): the checker for agz_train_step.  Parameters live as flat vectors in the Flux
layouts the C ABI uses (conv [kw,kh,cin,cout] column-major, Dense [out,in] column-major), so that updated
parameters compare directly with agz_net_get_weights.  Test infrastructure only."""
import numpy as np
import torch

DT = torch.float64
K_W, K_B, K_BETA, K_GAMMA, K_MEAN, K_VAR, K_EPS = range(7)
L_VCONV, L_PCONV, L_VFC1, L_VFC2, L_PFC = -1, -2, -3, -4, -5


class Twin:
    def __init__(self, N, tower, get, dtype=DT):
        """get(layer, kind) -> flat float32 numpy array (e.g. Engine.get_weights).  dtype = torch.float32 makes the
        twin a plain PyTorch fp32 autograd of the same step: what f32 arithmetic costs at this depth, the yardstick
        for the device step's own distance from the float64 twin (tests/test_gpu_train.py)"""
        DT = dtype
        self.dt = dtype
        self.N, self.tower = N, tower
        self.P, self.A = N * N, N * N + 1
        self.convs = list(range(1 + 2 * tower)) + [L_VCONV, L_PCONV]
        self.th, self.run, self.eps, self.vel = {}, {}, {}, {}
        for l in self.convs:
            for k in (K_W, K_B, K_BETA, K_GAMMA):
                self.th[(l, k)] = torch.tensor(get(l, k), dtype=DT, requires_grad=True)
            self.run[l] = [torch.tensor(get(l, K_MEAN), dtype=DT), torch.tensor(get(l, K_VAR), dtype=DT)]
            self.eps[l] = float(get(l, K_EPS)[0])
        for l in (L_VFC1, L_VFC2, L_PFC):
            for k in (K_W, K_B):
                self.th[(l, k)] = torch.tensor(get(l, k), dtype=DT, requires_grad=True)
        for key, t in self.th.items():
            self.vel[key] = torch.zeros_like(t)

    def _conv_bn(self, l, k, cin, cout, x, training):
        w = self.th[(l, K_W)].reshape(cout, cin, k, k).permute(0, 1, 3, 2)      # column-major [a,b,ci,o] -> [o,ci,a,b]
        w = torch.flip(w, dims=(2, 3))                                        # NNlib conv is a true convolution
        y = torch.nn.functional.conv2d(x, w, self.th[(l, K_B)], padding=k // 2)
        return torch.nn.functional.batch_norm(y, self.run[l][0], self.run[l][1], self.th[(l, K_GAMMA)], self.th[(l, K_BETA)],
                                              training=training, momentum=0.1,
                                              eps=max(self.eps[l], 1e-5) if training else self.eps[l])

    def forward(self, feats, training):
        N, P, A, B = self.N, self.P, self.A, feats.shape[0]
        x = torch.tensor(np.asarray(feats, np.float64).reshape(B, 17, N, N).transpose(0, 1, 3, 2).copy(), dtype=self.dt)   # [b,c,i,j]
        h = torch.relu(self._conv_bn(0, 3, 17, 256, x, training))
        for blk in range(self.tower):
            t = torch.relu(self._conv_bn(1 + 2 * blk, 3, 256, 256, h, training))
            h = torch.relu(self._conv_bn(2 + 2 * blk, 3, 256, 256, t, training) + h)
        vh = torch.relu(self._conv_bn(L_VCONV, 1, 256, 1, h, training))
        ph = torch.relu(self._conv_bn(L_PCONV, 1, 256, 2, h, training))
        vflat = vh.permute(0, 1, 3, 2).reshape(B, P)
        pflat = ph.permute(0, 1, 3, 2).reshape(B, 2 * P)
        w1 = self.th[(L_VFC1, K_W)].reshape(P, 256).T
        w2 = self.th[(L_VFC2, K_W)].reshape(256, 1).T
        wp = self.th[(L_PFC, K_W)].reshape(2 * P, A).T
        v = torch.tanh(torch.relu(vflat @ w1.T + self.th[(L_VFC1, K_B)]) @ w2.T + self.th[(L_VFC2, K_B)])[:, 0]
        logp = torch.log_softmax(pflat @ wp.T + self.th[(L_PFC, K_B)], dim=1)
        return logp, v

    def step(self, feats, pi, z, eta=0.02, rho=0.9):
        """returns (total, policy, value, reg) before the update"""
        B = feats.shape[0]
        logp, v = self.forward(feats, True)
        pi_t, z_t = torch.tensor(np.asarray(pi, np.float64), dtype=self.dt), torch.tensor(np.asarray(z, np.float64), dtype=self.dt)
        lp = 0.01 * (-(pi_t * logp).sum() / B)
        lv = 0.01 * ((v - z_t) ** 2).mean()
        lr = 1e-4 * sum((t ** 2).sum() for t in self.th.values())
        loss = lp + lv + lr
        grads = torch.autograd.grad(loss, list(self.th.values()))
        with torch.no_grad():
            for (key, t), g in zip(self.th.items(), grads):
                self.vel[key] = rho * self.vel[key] - eta * g
                t += self.vel[key]
        return float(loss.detach()), float(lp.detach()), float(lv.detach()), float(lr.detach())

    def param(self, layer, kind):
        if kind == K_MEAN:
            return self.run[layer][0].numpy()
        if kind == K_VAR:
            return self.run[layer][1].numpy()
        return self.th[(layer, kind)].detach().numpy()
