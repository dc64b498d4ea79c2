"""SequenceModeling stage operator: BidirectionalLSTM on the HIP path (reference modules/sequence_modeling.py:4-22).

nn.LSTM / nn.Linear are parameter containers (state_dict keys rnn.weight_ih_l0[_reverse] ... linear.weight).
Forward = one GEMM for both directions' input projections, one persistent recurrent launch, one output GEMM.
"""
import torch
import torch.nn as nn

from .. import ops
from ._nn import require_no_grad


class BidirectionalLSTM(nn.Module):
    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        self.rnn = nn.LSTM(input_size, hidden_size, bidirectional=True, batch_first=True)
        self.linear = nn.Linear(hidden_size * 2, output_size)
        self.hidden_size = hidden_size

    def _packed(self):
        r = self.rnn
        ps = (r.weight_ih_l0, r.weight_ih_l0_reverse, r.weight_hh_l0, r.weight_hh_l0_reverse, r.bias_ih_l0,
              r.bias_hh_l0, r.bias_ih_l0_reverse, r.bias_hh_l0_reverse)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cache = getattr(self, "_mrn_packed", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w_ih = torch.cat([ps[0], ps[1]], 0).contiguous()            # [2*4H, in]
                w_hh = torch.stack([ops.pack_fragment_major(ps[2]), ops.pack_fragment_major(ps[3])], 0).contiguous()
                b_ih = torch.cat([ps[4], ps[6]], 0).contiguous()
                b_hh = torch.cat([ps[5], ps[7]], 0).contiguous()
            cache = (key, (w_ih, w_hh, b_ih, b_hh))
            self._mrn_packed = cache
        return cache[1]

    def forward(self, input, out=None):
        """input [B,T,in] -> [B,T,out]; `out` may be a strided [B,T,out] view (zero-copy expert stacking)."""
        from ..functional import BiLSTMFn, LinearFn, needs_grad
        if needs_grad(self, input):
            r = self.rnn
            rec = BiLSTMFn.apply(input, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0, r.weight_ih_l0_reverse,
                                 r.weight_hh_l0_reverse, r.bias_ih_l0_reverse, r.bias_hh_l0_reverse)
            y = LinearFn.apply(rec, self.linear.weight, self.linear.bias)
            if out is not None:
                out.copy_(y)
                return out
            return y
        w_ih, w_hh, b_ih, b_hh = self._packed()
        xproj = ops.linear(input, w_ih, b_ih)
        if ops.RECURRENT_X3 and self.hidden_size == 256:
            # inference: the recurrent product as split-fp16 x3, as the lock-step groups run it (half the step time of the exact-fp32 MFMA)
            w_h, w_inv = self._packed_x3()
            rec = ops.lstm_layer_x3_grouped(xproj.unsqueeze(0), w_h, w_inv, b_hh.unsqueeze(0), self.hidden_size, 2)[0]
        else:
            rec = ops.lstm_layer(xproj, w_hh, b_hh, self.hidden_size, 2)
        return ops.linear(rec, self.linear.weight, self.linear.bias, out=out)

    def _packed_x3(self):
        r = self.rnn
        ps = (r.weight_hh_l0, r.weight_hh_l0_reverse)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cache = getattr(self, "_mrn_packed_x3", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                packs = [ops.pack_fragment_major_h(w) for w in ps]
                cache = (key, (torch.stack([p_[0] for p_ in packs]).unsqueeze(0).contiguous(),
                               torch.cat([p_[1] for p_ in packs]).unsqueeze(0).contiguous()))
            self._mrn_packed_x3 = cache
        return cache[1]
