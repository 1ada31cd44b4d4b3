"""BaseModel: device selection and checkpoint I/O with the reference's semantics (codes/models/base_model.py:11-190).

Checkpoint format: torch.save({'model_state_dict', 'optimizer_state_dict'}) as '<step>_<label>.pth'.  Loading is POSITIONAL
(i-th loaded tensor -> i-th current key), tolerates renamed keys, prefixes 'generated_image_model.' when the current generator is
CEM-wrapped, zero-extends the leading input channels of conv weights when a latent-input generator is initialised from a plain
ESRGAN checkpoint, and never overwrites the CEM's fixed filter taps.
"""
import collections
import os

import numpy as np
import torch
import torch.nn as nn

import CEM.CEMnet as CEMnet


class BaseModel():
    def __init__(self, opt):
        self.opt = opt
        self.save_dir = opt['path']['models']
        self.device = torch.device('cuda' if opt['gpu_ids'] is not None else 'cpu')
        self.is_train = opt['is_train']
        self.schedulers = []
        self.optimizers = []

    def feed_data(self, data):
        pass

    def optimize_parameters(self):
        pass

    def get_current_visuals(self):
        pass

    def get_current_losses(self):
        pass

    def print_network(self):
        pass

    def save(self, label):
        pass

    def load(self):
        pass

    def update_learning_rate(self, cur_step=None):
        for scheduler in self.schedulers:
            scheduler.step()

    def get_current_learning_rate(self):
        return self.schedulers[0].get_last_lr()[0] if self.schedulers else self.optimizers[0].param_groups[0]['lr']

    def get_network_description(self, network):
        if isinstance(network, nn.DataParallel):
            network = network.module
        return str(network), sum(p.numel() for p in network.parameters())

    def save_network(self, save_dir, network, network_label, iter_label, optimizer):
        save_path = os.path.join(save_dir, '{}_{}.pth'.format(iter_label, network_label))
        if isinstance(network, nn.DataParallel):
            network = network.module
        model_state_dict = collections.OrderedDict((k, v.cpu()) for k, v in network.state_dict().items())
        os.makedirs(save_dir, exist_ok=True)
        torch.save({'model_state_dict': model_state_dict, 'optimizer_state_dict': optimizer.state_dict()}, save_path)
        return save_path

    def load_network(self, load_path, network, strict=False, optimizer=None):
        if isinstance(network, nn.DataParallel):
            network = network.module
        loaded = torch.load(load_path, map_location='cpu')
        if 'optimizer_state_dict' in loaded.keys():
            if optimizer is not None:
                optimizer.load_state_dict(loaded['optimizer_state_dict'])
            loaded = loaded['model_state_dict']
        if self.opt['network_G']['CEM_arch']:
            loaded = CEMnet.Adjust_State_Dict_Keys(loaded, network.state_dict())
        loaded = self.process_loaded_state_dict(loaded_state_dict=loaded, current_state_dict=network.state_dict())
        network.load_state_dict(loaded, strict=strict)

    def Set_Require_Grad_Status(self, network, status):
        # (the CEM's fixed filter taps stay frozen: the reference flips them too, harmlessly — they are in no optimizer — but autograd then
        # spends a depth-wise weight gradient on them every step)
        cache = self.__dict__.setdefault('_trainable', {})
        entry = cache.get(id(network))
        if entry is None or entry[0] is not network:      # the module tree is walked once per network (702 parameters for RRDB-23)
            entry = cache[id(network)] = (network, [p for name, p in network.named_parameters() if 'Filter_OP' not in name])
        for p in entry[1]:
            p.requires_grad = status

    def process_loaded_state_dict(self, loaded_state_dict, current_state_dict):
        """Positional key matching + latent zero-extension (reference base_model.py:146-190)."""
        LATENT_WEIGHTS_RELATIVE_STD = 0.
        modified = collections.OrderedDict()
        current_keys = list(current_state_dict.keys())
        assert len(current_keys) == len(loaded_state_dict), 'Loaded model and current one should have the same number of parameters'
        renamed = extended = 0
        lat = getattr(self, 'num_latent_channels', 0) if getattr(self, 'latent_input', None) is not None else 0
        cem_ops = getattr(getattr(self, 'CEM_net', None), 'OP_names', []) if getattr(self, 'CEM_arch', False) else []
        for i, key in enumerate(loaded_state_dict.keys()):
            current_key = current_keys[i]
            loaded_t, current_t = loaded_state_dict[key], current_state_dict[current_key]
            if key != current_key:
                assert loaded_t.size()[:1] + loaded_t.size()[2:] == current_t.size()[:1] + current_t.size()[2:], \
                    'Unmatching parameter sizes after changing parameter key name'
                renamed += 1
            if lat > 0 and 'weight' in key and loaded_t.dim() > 1 and current_t.size(1) in list(loaded_t.size(1) + np.arange(lat) + 1):
                # new latent input channels come FIRST (architecture.py:300): prepend (zero-std) weights for them
                extra = current_t.size(1) - loaded_t.size(1)
                new_w = current_t[:, :extra].to(loaded_t.dtype).cpu()
                std = new_w.std()
                new_w = LATENT_WEIGHTS_RELATIVE_STD * loaded_t.std() / std * new_w if std > 0 else 0 * new_w
                modified[current_key] = torch.cat([new_w, loaded_t.cpu()], 1)
                extended += 1
            elif any(op in key for op in cem_ops):
                continue   # the CEM's fixed filter taps are never loaded
            else:
                modified[current_key] = loaded_t
        if renamed > 0:
            print('Warning: Modified %d key names due to the change to using ModuleLists' % renamed)
        if extended > 0:
            print('Warning: %d model weights were augmented with zeros to accommodate for larger inputs' % extended)
        return modified
