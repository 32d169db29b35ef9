"""Raw downloads -> PNG pairs of one size, without h5py / scipy.misc: the replacement of the reference's
tools/data_preprocessor.py (`make preprocess`).  Same interface: DATA_DIR, WIDTH, HEIGHT, DHEIGHT, DWIDTH, START, LIMIT,
FORCE in the environment, dataset names as arguments (none = all four).

    <DATA_DIR>/nyu/unpacked/nyu_depth_v2_labeled.mat                     ->  <DATA_DIR>/nyu/{train,test}/<name>-{image,depth}.png
    <DATA_DIR>/make3d1/unpacked/{Train400Img,Train400Depth,Test134,Test134Depth}/  ->  <DATA_DIR>/make3d1/{train,test}/...
    <DATA_DIR>/make3d2/unpacked/Dataset{3,2}_{Images,Depths}/            ->  <DATA_DIR>/make3d2/{train,test}/...
    <DATA_DIR>/mnist/unpacked/{train-*,t10k-*}                           ->  moved to <DATA_DIR>/mnist/{train,test}/

nyu, as the reference does it (tools/data_preprocessor.py:167-210): every sample's image is resized to WIDTH x HEIGHT and
its depth map to DWIDTH x DHEIGHT (scipy.misc.imresize: the depth map is min-max scaled to 8 bits PER IMAGE on the way —
the metric scale is gone after this step, and that is the value convention the training path inherits), both are
turned by 90 degrees clockwise, every fifth sample (c % 5 == 0) goes to test/, the others to train/, and the file name
is the sample's rawRgbFilenames entry with '/' and '.' replaced.  The .mat file is a MATLAB v7.3 = HDF5 file, read by
ann3depth_amd/hdf5.py; resizing and PNG writing are ann3depth_amd/imresize.py and png.py.

make3d1 / make3d2 (tools/data_preprocessor.py:65-143): JPEG images and level-5 .mat depth files (ann3depth_amd/matv5.py:
`Position3DGrid[..., 3]` / `depthMap`), paired by position in the two directory listings after the reference's filter
(names ending in txt / db are dropped) and named after the depth file (between its first '-' and first '.'); make3d2's
images are turned clockwise before resizing; a sample that raises ValueError is reported and skipped.  The reference
pairs the listings in os.listdir order, which is whatever the file system returns; this build sorts both listings, the
one order in which the Make3D names (img-<id>.jpg / depth_sph_corr-<id>.mat) pair up on every file system.  JPEG decoding
is Pillow's (the library scipy.misc.imread called); without Pillow these two processors stop with a message.
mnist (:146-164): the unpacked idx files are moved, nothing is converted.
"""
import os
import shutil
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import hdf5, imresize, matv5  # noqa: E402


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


def include(name):
    """tools/data_preprocessor.py:33-35: files that are not samples (extension txt / db; a name without '.' raises, as
    there)."""
    return name[name.index('.') + 1:] not in ['txt', 'db']


def imread(path):
    """scipy.misc.imread(path): PIL.Image.open + fromimage — palette images become RGB(A), bilevel ones 8-bit grey."""
    try:
        from PIL import Image
    except ImportError as e:                                                 # pragma: no cover - Pillow is in the image
        raise RuntimeError('the Make3D processors decode JPEG files with Pillow, which is not installed') from e
    with Image.open(path) as im:
        if im.mode == 'P':
            im = im.convert('RGBA' if 'transparency' in im.info else 'RGB')
        elif im.mode == '1':
            im = im.convert('L')
        return np.array(im)


def process_make3d(datadir, path_train, path_test, cfg, log, key, depth_dirs, img_dirs, depth_of, turn):
    log(f"Images: {cfg['width']}x{cfg['height']} Depths: {cfg['d_width']}x{cfg['d_height']}")
    path = os.path.join(datadir, key, 'unpacked')
    targets = [path_train, path_test]
    empty_dirs_or_fail(targets, cfg['force'])
    written = 0
    for dd, idir, tp in zip(depth_dirs, img_dirs, targets):
        dp, ip = os.path.join(path, dd), os.path.join(path, idir)
        log(f'Preprocessing images in {dp} and {ip}')
        depths = sorted(filter(include, os.listdir(dp)))
        imgs = sorted(filter(include, os.listdir(ip)))
        c = cfg['start']
        for d, i in zip(depths[cfg['start']:cfg['limit']], imgs[cfg['start']:cfg['limit']]):
            try:
                name = d[d.index('-') + 1:d.index('.')]
                img = imread(os.path.join(ip, i))
                if turn:
                    img = np.rot90(img, k=-1)
                img = imresize.imresize(img, (cfg['width'], cfg['height']))
                depth = depth_of(matv5.loadmat(os.path.join(dp, d)))
                depth = imresize.imresize(depth, (cfg['d_width'], cfg['d_height']))
            except ValueError as ve:
                log(f'Skipping sample {c}, {d} and {i}. Reason: {ve}')
                continue
            c += 1
            imresize.imsave(os.path.join(tp, f'{name}-image.png'), img)
            imresize.imsave(os.path.join(tp, f'{name}-depth.png'), depth)
            written += 1
    return written


def process_make3d1(datadir, path_train, path_test, cfg, log=print):
    """tools/data_preprocessor.py:65-104: depth = the fourth plane of Position3DGrid (55 x 305 x 4: the laser's range)."""
    return process_make3d(datadir, path_train, path_test, cfg, log, 'make3d1', ['Train400Depth', 'Test134Depth'],
                          ['Train400Img', 'Test134'], lambda mat: mat['Position3DGrid'][..., 3], turn=False)


def process_make3d2(datadir, path_train, path_test, cfg, log=print):
    """tools/data_preprocessor.py:107-143: Dataset3 trains, Dataset2 tests; the images lie on their side."""
    return process_make3d(datadir, path_train, path_test, cfg, log, 'make3d2', ['Dataset3_Depths', 'Dataset2_Depths'],
                          ['Dataset3_Images', 'Dataset2_Images'], lambda mat: mat['depthMap'], turn=True)


def process_mnist(datadir, path_train, path_test, cfg, log=print):
    """tools/data_preprocessor.py:146-164: train-* -> train/, t10k-* -> test/, anything else stays."""
    empty_dirs_or_fail([path_train, path_test], cfg['force'])
    path = os.path.join(datadir, 'mnist', 'unpacked')
    moved = 0
    for fn in sorted(os.listdir(path)):
        if fn.startswith('train-'):
            goal = path_train
        elif fn.startswith('t10k-'):
            goal = path_test
        else:
            log(f'Skipping {fn}')
            continue
        log(f'Moving {fn}')
        shutil.move(os.path.join(path, fn), goal)
        moved += 1
    return moved


PROCESSORS = {'make3d1': process_make3d1, 'make3d2': process_make3d2, 'nyu': process_nyu, 'mnist': process_mnist}


def main(argv=None, env=None, log=print):
    argv = sys.argv[1:] if argv is None else argv
    env = os.environ if env is None else env
    cfg = settings(env)
    log('\nPreprocessing data...')
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
