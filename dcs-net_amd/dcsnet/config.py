"""Hyper-parameters and layer geometry the hot path is built from (the reference keeps them in config.py:31-53 and :55-116).
The VALUES are the reference's; this module is organised by what consumes them, is importable without torchaudio /
complexPyTorch, and leaves out everything that touches the file system or the audio backend (data loading is out of scope:
SURVEY.md §8f).  `hparams` stays a plain dict and `config` a plain attribute bag because the reference's modules read them
by key / by attribute (c_network.py:102-163, network_functions.py:210-258)."""
import torch
from torch import nn

from .complexLayers import ComplexReLU
from .network_functions import SiSNR, wSDR, ComplexLReLU

_MODEL = dict(no_of_layers=7, channels=[1, 16, 32, 64, 128, 256, 256, 256], lstm_layers=2, lstm_bidir=True,
              channel_attention_reduction_ratio=16, spatial_attention_kernel_size=7,
              initialisation_distribution=nn.init.xavier_uniform_)
_REGULARISATION = dict(dropout=True, dropout_conv=0.1, dropout_fc=0.2, stochastic_weight_avg=True)
_OPTIMISER = dict(lr=10e-5, optim_eps=10e-7, optim_weight_decay=10e-5, optim_amsgrad=True, batch_size=32,
                  gradient_clip_val=100.0, gradient_clip_algorithm='norm')
_LOSS = dict(speech_alpha=0.7, noise_loss_type=6, speech_loss_type=0, atan2_eps=10e-7)
_DATA = dict(dataset_type=28)
hparams = {**_OPTIMISER, **_MODEL, **_LOSS, **_REGULARISATION, **_DATA}

_ENCODER_KERNELS = [7, 7, 5, 5, 3, 3, 3]                    # frequency x time, square
_DECODER_KERNELS = [3] * 7
_ENCODER_STRIDES = [(2, 2)] * 3 + [(2, 1)] * 4              # (frequency, time): time is halved three times only
_DECODER_UPSAMPLE = [(2, 1)] * 4 + [(2, 2)] * 3             # the mirror image of the strides


class Config(object):
    def __init__(self):
        gpus = torch.cuda.device_count() if torch.cuda.is_available() else 0
        sr, n_fft, hop = 16000, 512, 32
        attrs = dict(
            # run control
            tune=False, max_epochs=200, num_gpus=gpus, num_loader_workers=4 * gpus, precision=32, seed=0,
            detect_anomaly=True, val_log_sample_size=1,
            # audio / STFT front end
            sr=sr, file_sr=48000, fft_size=n_fft, window_length=n_fft, hop_length=hop,
            window=torch.hann_window(window_length=n_fft), normalise_audio=True, normalise_stft=True,
            # losses
            L1=nn.L1Loss(), mse=nn.MSELoss(), SiSNR=SiSNR(), wSDR=wSDR(),
            # layer geometry
            kernel_sizeE=list(_ENCODER_KERNELS), kernel_sizeD=list(_DECODER_KERNELS),
            paddingE=[k // 2 for k in _ENCODER_KERNELS], paddingD=[k // 2 for k in _DECODER_KERNELS],
            strideE=list(_ENCODER_STRIDES), strideD=(1, 1), upsample_scale_factor=list(_DECODER_UPSAMPLE),
            upsampling_mode='nearest',
            RactivationE=nn.ReLU, RactivationD=nn.LeakyReLU, CactivationE=ComplexReLU, CactivationD=ComplexLReLU,
            # derived figures the reference logs
            receptive_field_freq=291 * (sr / n_fft), receptive_field_time=291 / (sr / hop),
            integer_win_size=int(((1000 / (sr / hop)) * (n_fft / 2) / 1000) * sr))
        attrs['data_params'] = {'batch_size': hparams['batch_size'], 'shuffle': True,
                                'num_workers': attrs['num_loader_workers'], 'pin_memory': True}
        self.__dict__.update(attrs)


config = Config()
