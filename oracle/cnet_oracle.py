"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path.

CPU restatement (pure PyTorch, fp32) of the reference's complex U-Net wiring:
``C_NETWORK.__init__`` / ``forward`` (c_network.py:88-226), ``ComplexLSTM``
(c_network.py:12-51), ``ComplexChannelAttention`` (c_network.py:53-69) and
``ComplexSpatialAttention`` (c_network.py:71-84), with the geometry literals of
config.py:31-53 and config.py:83-106.

PARITY STATUS: the WIRING is pinned — ``oracle/make_golden.py`` imports the
reference's own c_network.py (with this repo's cpt_oracle standing in for the
absent complexPyTorch package and a stub for the absent trainer package) and
stores seeded input -> mask vectors in tests/golden/cnet_*.npz, which
tests/test_oracle.py compares against this file.  The layer arithmetic below
the wiring is cpt_oracle's: parity unpinned there (see its header).

Sub-module names and registration order follow the reference because they are
its checkpoint contract (state_dict keys).
"""
import torch
from torch import nn

from . import cpt_oracle as cpt
from . import nf_oracle as nf

# config.py:31-53 (only what the forward path reads)
HPARAMS = {
    'no_of_layers': 7,
    'channels': [1, 16, 32, 64, 128, 256, 256, 256],
    'lstm_layers': 2,
    'lstm_bidir': True,
    'dropout_conv': 0.1,
    'dropout_fc': 0.2,
    'atan2_eps': 10e-7,
    'channel_attention_reduction_ratio': 16,
    'spatial_attention_kernel_size': 7,
}
# config.py:83-106
KERNEL_E = [7, 7, 5, 5, 3, 3, 3]
STRIDE_E = [(2, 2), (2, 2), (2, 2), (2, 1), (2, 1), (2, 1), (2, 1)]
KERNEL_D = [3] * 7
UPSAMPLE = [(2, 1), (2, 1), (2, 1), (2, 1), (2, 2), (2, 2), (2, 2)]


class ComplexLSTM(nn.Module):
    """c_network.py:12-51 without the unused projection branch."""

    def __init__(self, input_size, hidden_size, num_layers, bidirectional):
        super().__init__()
        kw = dict(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                  bidirectional=bidirectional, batch_first=True)
        self.real_lstm = nn.LSTM(**kw)
        self.imag_lstm = nn.LSTM(**kw)

    def forward(self, x):
        re, im = x.real, x.imag
        rr = self.real_lstm(re)[0]
        ri = self.imag_lstm(re)[0]
        ir = self.real_lstm(im)[0]
        ii = self.imag_lstm(im)[0]
        return torch.complex(rr - ii, ir + ri)


class ComplexChannelAttention(nn.Module):
    """c_network.py:53-69 (the 'max' branch is an average: nf_oracle quirk)."""

    def __init__(self, channels, ratio):
        super().__init__()
        hidden = max(channels // ratio, 1)
        self.fc = nn.Sequential(cpt.ComplexConv2d(channels, hidden, kernel_size=1, bias=False),
                                cpt.ComplexReLU(),
                                cpt.ComplexConv2d(hidden, channels, kernel_size=1, bias=False))

    def forward(self, x):
        a = self.fc(nf.complex_adaptive_avg_pool2d(x, 1))
        m = self.fc(nf.complex_adaptive_max_pool2d(x, 1))
        return nf.complex_sigmoid(a + m)


class ComplexSpatialAttention(nn.Module):
    """c_network.py:71-84."""

    def __init__(self, k):
        super().__init__()
        self.conv1 = cpt.ComplexConv2d(2, 1, k, padding=k // 2, bias=False)

    def forward(self, x):
        mean_c = torch.mean(x, dim=1, keepdim=True)
        max_c = torch.complex(torch.max(x.real, dim=1, keepdim=True)[0],
                              torch.max(x.imag, dim=1, keepdim=True)[0])
        return nf.complex_sigmoid(self.conv1(torch.cat([mean_c, max_c], dim=1)))


class C_NETWORK_Oracle(nn.Module):
    """c_network.py:88-226.  ``hp`` overrides HPARAMS (tests pass dropout 0)."""

    def __init__(self, hp=None, init=nn.init.xavier_uniform_):
        super().__init__()
        self.hp = dict(HPARAMS)
        if hp:
            self.hp.update(hp)
        ch, L = self.hp['channels'], self.hp['no_of_layers']
        ratio, sk = self.hp['channel_attention_reduction_ratio'], self.hp['spatial_attention_kernel_size']
        # registration order of c_network.py:95-98
        self.encoder = nn.ModuleList()
        self.decoder = nn.ModuleList()
        self.decoder_attention = nn.ModuleList()
        self.skip_attention = nn.ModuleList()

        self.initial_batchnorm = cpt.ComplexBatchNorm2d(max(ch[0] // 2, 1))
        for i in range(L):
            cin = 1 if i == 0 else ch[i] // 2
            cout = ch[i + 1] // 2
            self.encoder.append(nn.Sequential(
                cpt.ComplexConv2d(cin, cout, KERNEL_E[i], STRIDE_E[i], KERNEL_E[i] // 2),
                cpt.ComplexBatchNorm2d(cout),
                cpt.ComplexReLU()))

        self.lstm = ComplexLSTM(ch[4], ch[4] // 2, self.hp['lstm_layers'], self.hp['lstm_bidir'])
        self.fc = cpt.ComplexLinear(ch[5] // 2, ch[5] // 2)

        for i in range(L):
            both = ch[L - i]            # complex channels of cat(d, skip)
            half = both // 2
            cout = max(ch[L - 1 - i] // 2, 1)
            convt = cpt.ComplexConvTranspose2d(both, cout, KERNEL_D[i], (1, 1), KERNEL_D[i] // 2)
            if i == L - 1:
                self.decoder.append(convt)
            else:
                self.decoder.append(nn.Sequential(convt, cpt.ComplexBatchNorm2d(cout), nf.ComplexLReLU()))
            self.skip_attention.append(ComplexChannelAttention(half, ratio))
            self.skip_attention.append(ComplexSpatialAttention(sk))
            self.decoder_attention.append(ComplexChannelAttention(cout, ratio))
            self.decoder_attention.append(ComplexSpatialAttention(sk))

        self.dropout_conv = nn.Dropout(self.hp['dropout_conv'])
        self.dropout_fc = nn.Dropout(self.hp['dropout_fc'])
        if init is not None:                     # c_network.py:174-184
            for m in self.modules():
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                    init(m.weight)

    @staticmethod
    def _drop(layer, z):
        return torch.view_as_complex(layer(torch.view_as_real(z)))

    def encode(self, x):
        feats = [self.initial_batchnorm(x.view(x.shape[0], -1, x.shape[1], x.shape[2]))]
        for blk in self.encoder:
            feats.append(self._drop(self.dropout_conv, blk(feats[-1])))
        return feats

    def latent(self, e):
        seq = torch.flatten(e, 2, 3).permute(0, 2, 1)
        z = self._drop(self.dropout_fc, self.fc(self.lstm(seq)))
        return z.permute(0, 2, 1).reshape(e.shape)

    def decode(self, d, feats):
        L = self.hp['no_of_layers']
        for i in range(L):
            skip = feats[L - i]
            skip = self.skip_attention[2 * i](skip) * skip
            skip = self.skip_attention[2 * i + 1](skip) * skip
            d = cpt.complex_upsample(torch.cat((d, skip), dim=1), scale_factor=UPSAMPLE[i], mode='nearest')
            d = self.decoder[i](d)
            if i != L - 1:                      # decoder_attention[12], [13] never run
                d = d * self.decoder_attention[2 * i](d)
                d = d * self.decoder_attention[2 * i + 1](d)
            d = self._drop(self.dropout_conv, d)
        return d

    def forward(self, x, return_intermediates=False):
        feats = self.encode(x)
        lat = self.latent(feats[-1])
        d = self.decode(lat, feats)
        out = nf.bound_cRM(torch.squeeze(d), self.hp)       # c_network.py:224-226
        if return_intermediates:
            return out, {'enc': feats, 'latent': lat, 'dec_raw': d}
        return out
