"""Raw downloads -> PNG pairs of one size, without h5py / scipy.misc: the replacement of the reference's
tools/data_preprocessor.py (`make preprocess`) for the NYU Depth v2 labelled set.  Same interface: DATA_DIR, WIDTH,
HEIGHT, DHEIGHT, DWIDTH, START, LIMIT, FORCE in the environment, dataset names as arguments.

    <DATA_DIR>/nyu/unpacked/nyu_depth_v2_labeled.mat  ->  <DATA_DIR>/nyu/{train,test}/<name>-image.png + <name>-depth.png

As the reference does it (tools/data_preprocessor.py:167-210): every sample's image is resized to WIDTH x HEIGHT and its
depth map to DWIDTH x DHEIGHT (scipy.misc.imresize: the depth map is min-max scaled to 8 bits PER IMAGE on the way —
the metric scale is gone after this step, and that is the value convention the training path inherits), both are
turned by 90 degrees clockwise, every fifth sample (c % 5 == 0) goes to test/, the others to train/, and the file name
is the sample's rawRgbFilenames entry with '/' and '.' replaced.  The .mat file is a MATLAB v7.3 = HDF5 file, read by
ann3depth_amd/hdf5.py; resizing and PNG writing are ann3depth_amd/imresize.py and png.py.

make3d1 / make3d2 (JPEG images + MATLAB v5 .mat depth maps) and mnist are not converted by this build: they need a JPEG
decoder and a v5 .mat reader, and the training path this build serves is benchmarked on NYU.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import hdf5, imresize  # noqa: E402


def settings(env=None):
    env = os.environ if env is None else env
    width, height = int(env.get('WIDTH', 640)), int(env.get('HEIGHT', 480))
    d_height = int(env.get('DHEIGHT', 55))
    d_width = int(env.get('DWIDTH', d_height * width // height))
    try:
        limit = int(env.get('LIMIT'))
    except (TypeError, ValueError):
        limit = None
    return {'width': width, 'height': height, 'd_width': d_width, 'd_height': d_height,
            'start': int(env.get('START', 0)), 'limit': limit, 'force': bool(env.get('FORCE'))}


def empty_dirs_or_fail(directories, force):
    """tools/data_preprocessor.py:39-62: with FORCE the directories are emptied, otherwise they must be empty."""
    for directory in directories:
        if force:
            for f in os.listdir(directory):
                os.remove(os.path.join(directory, f))
        elif os.listdir(directory):
            raise FileExistsError(f'Directory is not empty: {directory}, aborting... Use FORCE=1!')


def sample_name(mat, ref):
    """rawRgbFilenames entry -> file stem: the characters of the referenced uint16 array, '/' and '.' -> '_', minus the
    last four characters (the former extension: 'living_room_0012/r-1234.ppm' -> 'living_room_0012_r-1234')."""
    chars = mat[ref][:].T[0]
    return ''.join(map(chr, chars)).replace('/', '_').replace('.', '_')[:-4]


def process_nyu(datadir, path_train, path_test, cfg, log=print):
    log(f"Images: {cfg['width']}x{cfg['height']} Depths: {cfg['d_width']}x{cfg['d_height']}")
    targets = [path_train, path_test]
    train_images = 5
    path = os.path.join(datadir, 'nyu', 'unpacked', 'nyu_depth_v2_labeled.mat')
    empty_dirs_or_fail(targets, cfg['force'])
    written = 0
    with hdf5.File(path) as mat:
        depths, images, names = mat['depths'], mat['images'], mat['rawRgbFilenames'][0]
        n = min(len(depths), len(images), len(names))
        for c in range(cfg['start'], n):
            if cfg['limit'] and c >= cfg['limit']:
                break
            img = imresize.imresize(images[c], (cfg['width'], cfg['height']))
            img = np.rot90(img, k=-1)
            depth = imresize.imresize(depths[c], (cfg['d_width'], cfg['d_height']))
            depth = np.rot90(depth, k=-1)
            name = sample_name(mat, names[c])
            out = targets[0 if c % train_images else 1]
            imresize.imsave(os.path.join(out, f'{name}-image.png'), img)
            imresize.imsave(os.path.join(out, f'{name}-depth.png'), depth)
            written += 1
    return written


PROCESSORS = {'nyu': process_nyu}
NOT_CONVERTED = ('make3d1', 'make3d2', 'mnist')


def main(argv=None, env=None, log=print):
    argv = sys.argv[1:] if argv is None else argv
    env = os.environ if env is None else env
    cfg = settings(env)
    log('\nPreprocessing data...')
    for key in NOT_CONVERTED:
        if key in argv:
            log(f'{key}: not converted by this build (see the module docstring)')
    for key, processor in PROCESSORS.items():
        if key not in argv and argv:
            continue
        train, test = os.path.join(env['DATA_DIR'], key, 'train'), os.path.join(env['DATA_DIR'], key, 'test')
        os.makedirs(train, 0o755, exist_ok=True)
        os.makedirs(test, 0o755, exist_ok=True)
        log(f'Preprocessing {key}')
        try:
            processor(env['DATA_DIR'], train, test, cfg, log)
        except FileExistsError as fe:
            log(fe)
    log('Preprocessing done.')
    return 0


if __name__ == '__main__':
    sys.exit(main())
