"""Hyper-parameters and layer geometry of the reference (config.py:31-53, :55-116), importable
without torchaudio / complexPyTorch.  Values are the reference's literals; only the pieces that
touch the file system or the audio backend are left out (data loading is out of scope:
SURVEY.md §8f)."""
import torch
from torch import nn

from .complexLayers import ComplexReLU
from .network_functions import SiSNR, wSDR, ComplexLReLU

hparams = {'lr': 10e-5,
           'initialisation_distribution': nn.init.xavier_uniform_,
           'speech_alpha': 0.7,
           'no_of_layers': 7,
           'channels': [1, 16, 32, 64, 128, 256, 256, 256],
           'lstm_layers': 2,
           'lstm_bidir': True,
           'noise_loss_type': 6,
           'speech_loss_type': 0,
           'dropout': True,
           'dropout_conv': 0.1,
           'dropout_fc': 0.2,
           'batch_size': 32,
           'optim_eps': 10e-7,
           'atan2_eps': 10e-7,
           'optim_weight_decay': 10e-5,
           'optim_amsgrad': True,
           'gradient_clip_val': 100.0,
           'gradient_clip_algorithm': "norm",
           'stochastic_weight_avg': True,
           'dataset_type': 28,
           'channel_attention_reduction_ratio': 16,
           'spatial_attention_kernel_size': 7}


class Config(object):
    def __init__(self):
        self.tune = False
        self.sr = 16000
        self.file_sr = 48000
        self.max_epochs = 200
        self.num_gpus = torch.cuda.device_count() if torch.cuda.is_available() else 0
        self.num_loader_workers = self.num_gpus * 4
        self.data_params = {'batch_size': hparams['batch_size'], 'shuffle': True,
                            'num_workers': self.num_loader_workers, 'pin_memory': True}
        self.precision = 32
        self.fft_size = 512
        self.window_length = self.fft_size
        self.hop_length = 32
        self.window = torch.hann_window(window_length=self.window_length)
        self.normalise_audio = True
        self.normalise_stft = True
        self.L1 = nn.L1Loss()
        self.mse = nn.MSELoss()
        self.SiSNR = SiSNR()
        self.wSDR = wSDR()
        self.kernel_sizeE = [7, 7, 5, 5, 3, 3, 3]
        self.kernel_sizeD = [3, 3, 3, 3, 3, 3, 3]
        self.paddingE = [k // 2 for k in self.kernel_sizeE]
        self.paddingD = [k // 2 for k in self.kernel_sizeD]
        self.strideE = [(2, 2), (2, 2), (2, 2), (2, 1), (2, 1), (2, 1), (2, 1)]
        self.strideD = (1, 1)
        self.RactivationE = nn.ReLU
        self.RactivationD = nn.LeakyReLU
        self.CactivationE = ComplexReLU
        self.CactivationD = ComplexLReLU
        self.upsample_scale_factor = [(2, 1), (2, 1), (2, 1), (2, 1), (2, 2), (2, 2), (2, 2)]
        self.upsampling_mode = 'nearest'
        self.receptive_field_freq = 291 * (self.sr / self.fft_size)
        self.receptive_field_time = 291 / (self.sr / self.hop_length)
        self.integer_win_size = int(((1000 / (self.sr / self.hop_length))
                                     * (self.window_length / 2) / 1000) * self.sr)
        self.val_log_sample_size = 1
        self.seed = 0
        self.detect_anomaly = True


config = Config()
