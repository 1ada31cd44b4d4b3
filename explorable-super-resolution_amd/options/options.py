"""Option files: JSON with `//` comments -> nested dict -> NoneDict, with the path / batch-size derivations the drivers rely on.
Same entry points as the reference's codes/options/options.py (parse, save, dict_to_nonedict, NoneDict); the explorable-JPEG
branches and the machine-specific dataset-root rewriting of the reference are not reproduced (outside the RRDB+CEM path).
"""
import json
import os
from collections import OrderedDict
from datetime import datetime

OVERRIDING_KEYS = [['train', 'resume'], ['datasets', 'train', 'n_workers'], ['train', 'val_running_avg_steps']]


def get_timestamp():
    return datetime.now().strftime('%y%m%d-%H%M%S')


def _strip_comments(path):
    with open(path, 'r') as f:
        return ''.join(line.split('//')[0] + '\n' for line in f)


def dictionary_values_choice(dictionary, chosen_option):
    """Option files may give a value per training phase ({"PhaseInit": a, "PhaseGAN": b}); pick `chosen_option` wherever it is a
    key; the string "None" means None (reference options.py:46-54)."""
    while isinstance(dictionary, dict) and chosen_option in dictionary.keys():
        dictionary = dictionary[chosen_option]
        if dictionary == "None":
            return None
    if isinstance(dictionary, dict):
        for key, value in dictionary.items():
            dictionary[key] = dictionary_values_choice(value, chosen_option)
    return dictionary


def _diff(a, b, prefix=''):
    """Minimal stand-in for DeepDiff: yields (path, old, new) for leaves that differ."""
    if isinstance(a, dict) and isinstance(b, dict):
        for k in sorted(set(a) | set(b), key=str):
            yield from _diff(a.get(k, '<absent>'), b.get(k, '<absent>'), '%s[%r]' % (prefix, k))
    elif a != b:
        yield prefix, a, b


def parse(opt_path, is_train=True, batch_size_multiplier=None, **kwargs):
    opt = parse_conf(opt_path=opt_path, is_train=is_train, batch_size_multiplier=batch_size_multiplier, **kwargs)
    if is_train and opt['train'].get('resume'):
        saved = parse_conf(opt_path=os.path.join(opt['path']['experiments_root'], 'options.json'), is_train=is_train,
                           batch_size_multiplier=batch_size_multiplier, **kwargs)
        for key in OVERRIDING_KEYS:
            cur, cur_saved = opt, saved
            for sub in key[:-1]:
                cur, cur_saved = cur[sub], cur_saved[sub]
            if key[-1] in cur:
                cur_saved[key[-1]] = cur[key[-1]]
        saved['train']['resume'] = opt['train']['resume']
        changes = [c for c in _diff(opt, saved) if 'timestamp' not in c[0]]
        if changes:
            print('Using some saved configuration values that are different from the current ones. This means changing:')
            for path, old, new in changes:
                print('%s:\n\tFrom: %s\n\tTo: %s' % (path, old, new))
        return saved
    return opt


def parse_conf(opt_path, is_train=True, batch_size_multiplier=None, **kwargs):
    name = kwargs.get('name')
    if kwargs.get('JPEG'):
        raise NotImplementedError('explorable JPEG decoding options are outside the RRDB+CEM path')
    opt = json.loads(_strip_comments(opt_path), object_pairs_hook=OrderedDict)
    opt = dictionary_values_choice(opt, 'PhaseInit' if kwargs.get('initialization') else 'PhaseGAN')
    scale = opt['scale']
    opt['timestamp'] = get_timestamp()
    opt['is_train'] = is_train
    if 'datasets' in opt:
        root = opt['path']['datasets'] if 'datasets' in opt['path'] else opt['path']['root']
        for phase, dataset in opt['datasets'].items():
            phase = phase.split('_')[0]
            dataset['phase'] = phase
            dataset['scale'] = scale
            is_lmdb = False
            for field in ('dataroot_HR', 'dataroot_LR'):
                if dataset.get(field) is not None:
                    dataset[field] = os.path.expanduser(os.path.join(root, dataset[field]))
                    is_lmdb = is_lmdb or dataset[field].endswith('lmdb')
            dataset['data_type'] = 'lmdb' if is_lmdb else 'img'
            if 'train' in opt and any(f in opt['train'] for f in ('pixel_domain', 'feature_domain')):
                assert opt['model'] in ['srragan', 'srgan'], 'Unsupported'
            if phase == 'train' and dataset.get('subset_file') is not None:
                dataset['subset_file'] = os.path.expanduser(dataset['subset_file'])
    for key, path in opt['path'].items():
        if path:
            opt['path'][key] = os.path.expanduser(path)
    if name is not None:
        opt['name'] = os.path.join(name)
    experiments_root = os.path.join(opt['path']['root'], 'experiments', opt['name'])
    opt['path']['experiments_root'] = experiments_root
    opt['path']['models'] = os.path.join(experiments_root, 'models')
    netG = opt['network_G']
    netG.setdefault('latent_input', 'None')
    if netG['latent_input'] == 'None':
        netG['latent_channels'] = 0
    netG.setdefault('padding', 1)
    opt['path']['log'] = experiments_root
    if is_train:
        opt['path']['val_images'] = os.path.join(experiments_root, 'val_images')
        tr_data, tr = opt['datasets']['train'], opt['train']
        if 'batch_size_per_GPU' not in tr_data:      # legacy files only give batch_size
            tr_data['batch_size_per_GPU'] = 1 * tr_data['batch_size']
        tr.setdefault('D_update_measure', 'post_train_D_diff')
        tr_data['batch_size'] = 1 * tr_data['batch_size_per_GPU']
        if batch_size_multiplier is not None:
            # the reference multiplies by the number of GPUs that nn.DataParallel splits the batch over (train.py:30); with one
            # process per GPU the multiplier is the world size and every rank then takes batch_size / world of it
            tr_data['batch_size'] *= batch_size_multiplier
            tr_data['n_workers'] *= batch_size_multiplier
        if 'batch_size_4_grads_G' not in tr_data:
            tr_data['batch_size_4_grads_G'] = tr_data['batch_size_4_grads_D'] = 1 * tr_data['batch_size']
        while tr_data['batch_size_4_grads_G'] % tr_data['batch_size'] or tr_data['batch_size_4_grads_D'] % tr_data['batch_size']:
            tr_data['batch_size'] -= 1
        assert tr_data['batch_size'] > 0, 'Batch size must be greater than 0'
        assert tr_data['batch_size_4_grads_D'] >= tr_data['batch_size_4_grads_G'], 'Currently not supporting G_batch>D_batch'
        tr['grad_accumulation_steps_G'] = tr_data['batch_size_4_grads_G'] // tr_data['batch_size']
        tr['grad_accumulation_steps_D'] = tr_data['batch_size_4_grads_D'] // tr_data['batch_size']
        if 'network_D' in opt and opt['network_D'] is not None:
            if opt['network_D']['which_model_D'] == 'PatchGAN':
                assert tr['gan_type'] in ['lsgan', 'wgan-gp', 'wgan-sn', 'wgan-sngp']
            else:
                assert tr['gan_type'] != 'lsgan', 'lsgan GAN type should be used with Patch discriminator. For regular D, use vanilla type.'
    else:
        opt['path']['results_root'] = os.path.join(opt['path']['root'], 'results', opt['name'])
    netG['scale'] = scale
    return opt


def save(opt):
    dump_dir = opt['path']['experiments_root'] if opt['is_train'] else opt['path']['results_root']
    os.makedirs(dump_dir, exist_ok=True)
    with open(os.path.join(dump_dir, 'options.json'), 'w') as dump_file:
        json.dump(opt, dump_file, indent=2)


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{key: dict_to_nonedict(sub) for key, sub in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(sub) for sub in opt]
    return opt
