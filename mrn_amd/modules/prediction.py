"""Prediction stage operator: attention decoder on the HIP path (reference modules/prediction.py:8-118).

Same constructor (`Attention(input_size, hidden_size, num_class, fc, num_char_embeddings=256)`), forward signature
and state_dict keys (attention_cell.{i2h,h2h,score,rnn}.*, generator.* aliasing fc, char_embeddings.weight).
Teacher-forced mode is one persistent kernel for all 26 steps with i2h(H) and the embedding half of the LSTMCell
input projection hoisted into GEMMs; greedy mode runs the same kernel step by step with the argmax fed back.
"""
import torch
import torch.nn as nn

from .. import ops
from ._nn import require_no_grad


class AttentionCell(nn.Module):
    def __init__(self, input_size, hidden_size, num_embeddings):
        super().__init__()
        self.i2h = nn.Linear(input_size, hidden_size, bias=False)
        self.h2h = nn.Linear(hidden_size, hidden_size)
        self.score = nn.Linear(hidden_size, 1, bias=False)
        self.rnn = nn.LSTMCell(input_size + num_embeddings, hidden_size)
        self.hidden_size = hidden_size
        self.input_size = input_size


class Attention(nn.Module):
    def __init__(self, input_size, hidden_size, num_class, fc, num_char_embeddings=256):
        super().__init__()
        self.attention_cell = AttentionCell(input_size, hidden_size, num_char_embeddings)
        self.hidden_size = hidden_size
        self.num_class = num_class
        self.generator = fc
        self.num_char_embeddings = num_char_embeddings
        self.char_embeddings = nn.Embedding(num_class, num_char_embeddings)

    def cut_unknown(self, index):
        return torch.where(index >= self.num_class, 0, index)

    def _packed(self):
        """fragment-major copies of the three recurrent weight streams (rebuilt when a parameter changes)"""
        cell = self.attention_cell
        ps = (cell.h2h.weight, cell.rnn.weight_ih, cell.rnn.weight_hh)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cache = getattr(self, "_mrn_packed", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                D = cell.input_size
                packed = (ops.pack_fragment_major(ps[0]), ops.pack_fragment_major(ps[1][:, :D]),
                          ops.pack_fragment_major(ps[2]))
            cache = (key, packed)
            self._mrn_packed = cache
        return cache[1]

    def _packed_x3(self):
        """the same three streams as fp16 hi / lo fragment-major splits + their inverse prescales (ops.pack_decoder_x3), cached"""
        cell = self.attention_cell
        ps = (cell.h2h.weight, cell.rnn.weight_ih, cell.rnn.weight_hh)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cache = getattr(self, "_mrn_packed_x3", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, ops.pack_decoder_x3(ps[0], ps[1], ps[2], cell.input_size))
            self._mrn_packed_x3 = cache
        return cache[1]

    def x3_ok(self):
        return ops.DECODER_X3 and self.attention_cell.input_size % 32 == 0 and self.hidden_size == 256

    def _decode(self, batch_H, Hproj, eproj, hid=None, h=None, c=None):
        cell = self.attention_cell
        if self.x3_ok():
            w_h2h, w_ih_ctx, w_hh, w_inv = self._packed_x3()
        else:
            (w_h2h, w_ih_ctx, w_hh), w_inv = self._packed(), None
        return ops.attn_decoder(batch_H, Hproj, eproj, w_h2h, cell.h2h.bias, cell.score.weight,
                                w_ih_ctx, w_hh, cell.rnn.bias_hh, self.hidden_size, hid=hid, h_state=h, c_state=c, w_inv=w_inv)

    def forward(self, batch_H, text, is_train=True, batch_max_length=25, out=None):
        """batch_H [B,T,D]; text [B,S] (teacher forcing) or [B] of [SOS] (greedy) -> logits [B,S,num_class].
        `out` may be a preallocated (possibly strided) [B,S,num_class] buffer."""
        from ..functional import AttnDecoderFn, needs_grad
        cell = self.attention_cell
        B = batch_H.shape[0]
        S = batch_max_length + 1
        if needs_grad(self, batch_H):
            if not is_train:
                raise NotImplementedError("greedy decoding is inference only; call it under torch.no_grad()")
            probs = AttnDecoderFn.apply(batch_H, text, cell.i2h.weight, cell.h2h.weight, cell.h2h.bias, cell.score.weight,
                                        cell.rnn.weight_ih, cell.rnn.weight_hh, cell.rnn.bias_ih, cell.rnn.bias_hh,
                                        self.char_embeddings.weight, self.generator.weight, self.generator.bias, S)
            if out is not None:
                out.copy_(probs)
                return out
            return probs
        D = cell.input_size
        batch_H = batch_H.contiguous()
        Hproj = ops.linear(batch_H, cell.i2h.weight)
        w_emb = cell.rnn.weight_ih[:, D:]                      # [4H, E] strided view
        if is_train:
            emb = ops.embed_gather(text[:, :S], self.char_embeddings.weight, self.num_class)
            eproj = ops.linear(emb, w_emb, cell.rnn.bias_ih)
            hid = self._decode(batch_H, Hproj, eproj)
            return ops.linear(hid, self.generator.weight, self.generator.bias, out=out)
        # greedy decode (reference :70-86): token_{s+1} = argmax(generator(h_s))
        dev = batch_H.device
        targets = text[0].expand(B).contiguous().view(B, 1)
        probs = out if out is not None else torch.empty(B, S, self.num_class, device=dev, dtype=torch.float32)
        h = torch.zeros(B, self.hidden_size, device=dev)
        c = torch.zeros(B, self.hidden_size, device=dev)
        hid = torch.empty(B, 1, self.hidden_size, device=dev)
        for s in range(S):
            emb = ops.embed_gather(targets, self.char_embeddings.weight, self.num_class)
            eproj = ops.linear(emb, w_emb, cell.rnn.bias_ih)
            self._decode(batch_H, Hproj, eproj, hid=hid, h=h, c=c)
            step = probs[:, s:s + 1, :]
            ops.linear(hid, self.generator.weight, self.generator.bias, out=step)
            targets = ops.argmax_lastdim(step).view(B, 1)
        return probs
