"""PCCModel (reference pcc_model.py:8-45): container wiring Encoder [1,16,32,64,32,8], Decoder [8,64,32,16] and
EntropyBottleneck(8).  Only the inference surface is on the encode/decode path; `forward` / `get_likelihood` are
training-only in the reference and are not provided."""
import torch

from .autoencoder import Encoder, Decoder
from .entropy_model import EntropyBottleneck


class PCCModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder(channels=[1, 16, 32, 64, 32, 8])
        self.decoder = Decoder(channels=[8, 64, 32, 16])
        self.entropy_bottleneck = EntropyBottleneck(8)

    def load_state_dict(self, state_dict, strict=True, **kw):
        from . import conventions
        self.entropy_bottleneck.invalidate()
        return super().load_state_dict(conventions.permute_state_dict(state_dict), strict=strict, **kw)

    def forward(self, x, training=True):
        raise NotImplementedError('PCCModel.forward is the training graph (pcc_model.py:26-45); use coder.Coder for encode/decode')


if __name__ == '__main__':
    print(PCCModel())
