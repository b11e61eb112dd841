"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path.

CPU restatement of the reference's real-valued twin ``R_NETWORK`` (DR-Net / DRS-Net, r_network.py:8-173):
BASELINE.json configs[0] ("1-utterance batch, CPU PyTorch reference, plumbing, no GPU").  It is out of scope
for kernels; it exists so that config 1 has a parity case.

PARITY STATUS: pinned — stock torch.nn layers only, and ``oracle/make_golden.py`` imports the reference's own
r_network.py and stores seeded input -> output in tests/golden/rnet_vectors.npz (tests/test_oracle.py).

Quirks kept: the channel attention returns sigmoid(fc(max_pool)) only — the avg branch is computed and then
overwritten (r_network.py:23-24); dropout_fc is gated by hparams['dropout'] (r_network.py:152) while
dropout_conv is not; ``torch.squeeze`` drops the batch dimension at B = 1 (r_network.py:171).
"""
import torch
from torch import nn

from .cnet_oracle import KERNEL_E, STRIDE_E, KERNEL_D, UPSAMPLE

CHANNELS = [1, 16, 32, 64, 128, 256, 256, 256]


class RealChannelAttention(nn.Module):
    def __init__(self, channels, ratio):
        super().__init__()
        hidden = max(channels // ratio, 1)
        self.fc = nn.Sequential(nn.Conv2d(channels, hidden, 1, bias=False), nn.ReLU(),
                                nn.Conv2d(hidden, channels, 1, bias=False))

    def forward(self, x):
        # AdaptiveMaxPool2d(1) (r_network.py:12,21): its gradient goes to the first maximum (torch.amax would share it among ties)
        return torch.sigmoid(self.fc(nn.functional.adaptive_max_pool2d(x, 1)))


class RealSpatialAttention(nn.Module):
    def __init__(self, k):
        super().__init__()
        self.conv1 = nn.Conv2d(2, 1, k, padding=k // 2, bias=False)

    def forward(self, x):
        pooled = torch.cat([x.mean(dim=1, keepdim=True), x.max(dim=1, keepdim=True)[0]], dim=1)
        return torch.sigmoid(self.conv1(pooled))


class R_NETWORK_Oracle(nn.Module):
    def __init__(self, dropout_conv=0.1, dropout_fc=0.2, dropout=True, ratio=16, sk=7):
        super().__init__()
        ch, L = CHANNELS, 7
        self.use_fc_dropout = dropout
        self.encoder, self.decoder = nn.ModuleList(), nn.ModuleList()
        self.decoder_attention, self.skip_attention = nn.ModuleList(), nn.ModuleList()
        self.initial_batchnorm = nn.BatchNorm2d(ch[0])
        for i in range(L):
            self.encoder.append(nn.Sequential(
                nn.Conv2d(1 if i == 0 else ch[i], ch[i + 1], KERNEL_E[i], STRIDE_E[i], KERNEL_E[i] // 2),
                nn.BatchNorm2d(ch[i + 1]), nn.ReLU()))
        self.lstm = nn.LSTM(input_size=ch[5], hidden_size=ch[4], num_layers=2, bidirectional=True, batch_first=True)
        self.fc = nn.Linear(ch[5], ch[5])
        self.dropout_conv, self.dropout_fc = nn.Dropout(dropout_conv), nn.Dropout(dropout_fc)
        for i in range(L):
            cin, cout = ch[L - i], max(ch[L - 1 - i], 1)
            convt = nn.ConvTranspose2d(2 * cin, cout, KERNEL_D[i], (1, 1), KERNEL_D[i] // 2)
            self.decoder.append(convt if i == L - 1 else nn.Sequential(convt, nn.BatchNorm2d(cout), nn.LeakyReLU()))
            self.skip_attention.append(RealChannelAttention(cin, ratio))
            self.skip_attention.append(RealSpatialAttention(sk))
            self.decoder_attention.append(RealChannelAttention(cout, ratio))
            self.decoder_attention.append(RealSpatialAttention(sk))

    def forward(self, x):
        L = 7
        feats = [self.initial_batchnorm(x.view(x.shape[0], -1, x.shape[1], x.shape[2]))]
        for blk in self.encoder:
            feats.append(self.dropout_conv(blk(feats[-1])))
        e = feats[-1]
        z = self.fc(self.lstm(torch.flatten(e, 2, 3).permute(0, 2, 1))[0])
        if self.use_fc_dropout:
            z = self.dropout_fc(z)
        d = z.permute(0, 2, 1).reshape(e.shape)
        for i in range(L):
            skip = feats[L - i]
            skip = self.skip_attention[2 * i](skip) * skip
            skip = self.skip_attention[2 * i + 1](skip) * skip
            d = nn.functional.interpolate(torch.cat((d, skip), dim=1), scale_factor=UPSAMPLE[i], mode='nearest')
            d = self.decoder[i](d)
            if i != L - 1:
                d = d * self.decoder_attention[2 * i](d)
                d = d * self.decoder_attention[2 * i + 1](d)
            d = self.dropout_conv(d)
        return torch.sigmoid(torch.squeeze(d))
