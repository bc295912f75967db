"""On-disk dataset of the inference scripts: `ImageFolderDataset` as `eval_seq.py:106-128,204` uses it.

Mirrors the reader side of the reference's `training_avatar_texture/dataset_new.py` (:29-195 `Dataset`, :197-367
`ImageFolderDataset`): constructor arguments, `__getitem__` / `get_by_name` / `get_label` / `get_vert` / `get_image` /
`get_uvImg` results and the shape / label properties.  The directory layout it reads:

    <root>/images512x512/dataset_realcam.json         {"labels": [[ "<video>/<frame>.png", [25 floats] ], ...]}  (camera per frame)
    <root>/images512x512/<label_file>                 same shape: the conditioning label per frame
    <root>/images512x512/<video>/<frame>.png          RGB(A) uint8
    <root>/orthRender256x256_face_eye/<video>/<frame>.npy   [256,256,>=3]: u, v in [-1,1], face mask (binarised at 0.5 here)
    <root>/coeffs/<video>/<frame>.npy                 FaceVerse coefficients (when `fvcoeffs_path` is given)
    <root>/uvRender256x256/<video>/<frame>.npy        [256,256,4] projected-vertex UV image, last channel masked (load_uv)
    <root>/uvRender256x256/<video>/<frame>_uvgttex.png  RGB texture in UV space (load_uv)
    <root>/fgmasks512x512/..., lms_counter512x512_newF/...   optional extra image channels (load_bg / load_lms_counter)

Zip archives, x-flips and `max_size` sub-sampling are training-set features; x-flip and max_size are kept (they are a few lines
of index arithmetic), zip reading is not (`_type` is always 'dir' in the reference as well, :220).  `cv2.imread` of the UV face
mask is done with PIL (OpenCV is not a dependency of this package); the mask is a single-channel PNG so the values agree."""
import json
import os

import numpy as np
import PIL.Image
import torch

from .. import dnnlib

UV_FACE_MASK = 'data_preprocess/FaceVerse/v3/dense_uv_expanded_mask_onlyFace.png'    # relative to the working directory (:228)


def _read_image_chw(path, resolution=None):
    with open(path, 'rb') as f:
        image = PIL.Image.open(f)
        if resolution:
            image = image.resize((resolution, resolution))
        image = np.array(image)
    if image.ndim == 2:
        image = image[:, :, None]
    return image.transpose(2, 0, 1)


def _binarise_mask_image(path, resolution=None):
    """First channel of a mask image as {127.5 -> stored as 127 in uint8, 255} (:311-313; the in-place uint8 assignment of the
    reference truncates 127.5 to 127)."""
    m = _read_image_chw(path, resolution)[:1]
    m[m > 127] = 255
    m[m < 128] = 127.5
    return m


class Dataset(torch.utils.data.Dataset):
    def __init__(self, name, raw_shape, max_size=None, use_labels=True, xflip=False, load_obj=True, return_name=False, random_seed=0):
        self._name = name
        self._raw_shape = list(raw_shape)
        self._use_labels = use_labels
        self._raw_labels = None
        self._label_shape = None
        self.load_obj = load_obj
        self.return_name = return_name
        idx = np.arange(self._raw_shape[0], dtype=np.int64)
        if max_size is not None and idx.size > max_size:
            np.random.RandomState(random_seed).shuffle(idx)
            idx = np.sort(idx[:max_size])
        self._xflip = np.zeros(idx.size, dtype=np.uint8)
        if xflip:
            idx = np.tile(idx, 2)
            self._xflip = np.concatenate([self._xflip, np.ones_like(self._xflip)])
        self._raw_idx = idx

    # -- to be provided by the concrete dataset
    def _load_raw_image(self, raw_idx, resolution=None):
        raise NotImplementedError

    def _load_raw_labels(self):
        raise NotImplementedError

    def get_label(self, idx):
        raise NotImplementedError

    def get_vert(self, idx):
        raise NotImplementedError

    def close(self):
        pass

    def _get_raw_labels(self):
        if self._raw_labels is None:
            labels = self._load_raw_labels() if self._use_labels else None
            if labels is None:
                labels = np.zeros([self._raw_shape[0], 0], dtype=np.float32)
            assert isinstance(labels, np.ndarray) and labels.shape[0] == self._raw_shape[0]
            assert labels.dtype in [np.float32, np.int64]
            if labels.dtype == np.int64:
                assert labels.ndim == 1 and np.all(labels >= 0)
            self._raw_labels = labels
            self._raw_labels_std = labels.std(0)
        return self._raw_labels

    def __getstate__(self):
        return dict(self.__dict__, _raw_labels=None)

    def __len__(self):
        return self._raw_idx.size

    def __getitem__(self, idx):
        raw = self._raw_idx[idx]
        image = self._load_raw_image(raw, resolution=self.resolution)
        if self._xflip[idx]:
            assert image.ndim == 3
            image = image[:, :, ::-1]
        item = (image.copy(), self.get_label(idx), self.get_vert(raw))
        return (self._image_fnames[raw],) + item if self.return_name else item

    def get_by_name(self, name):
        raw = self._image_fnames.index(name)
        image = self._load_raw_image(raw, resolution=self.resolution)
        label_cam = np.concatenate([self._get_raw_labels()[raw], self._raw_cams[raw]], axis=-1)
        return image.copy(), label_cam, self.get_vert(raw)

    def get_details(self, idx):
        d = dnnlib.EasyDict()
        d.raw_idx = int(self._raw_idx[idx])
        d.xflip = int(self._xflip[idx]) != 0
        d.raw_label = self._get_raw_labels()[d.raw_idx].copy()
        return d

    def get_label_std(self):
        return self._raw_labels_std

    @property
    def name(self):
        return self._name

    @property
    def image_shape(self):
        return list(self._raw_shape[1:])

    @property
    def num_channels(self):
        assert len(self.image_shape) == 3
        return self.image_shape[0]

    @property
    def resolution(self):
        assert len(self.image_shape) == 3 and self.image_shape[1] == self.image_shape[2]
        return self.image_shape[1]

    @property
    def label_shape(self):
        if self._label_shape is None:
            labels = self._get_raw_labels()
            self._label_shape = [int(np.max(labels)) + 1] if labels.dtype == np.int64 else labels.shape[1:]
        return list(self._label_shape)

    @property
    def label_dim(self):
        assert len(self.label_shape) == 1
        return self.label_shape[0]

    @property
    def has_labels(self):
        return any(x != 0 for x in self.label_shape)

    @property
    def has_onehot_labels(self):
        return self._get_raw_labels().dtype == np.int64


class ImageFolderDataset(Dataset):
    def __init__(self, path, mesh_path=None, mesh_type='.obj', resolution=None, load_exp=False, load_lms_counter=False, load_uv=False,
                 load_bg=False, label_file='dataset.json', fvcoeffs_path=None, **super_kwargs):
        self._path, self._mesh_path, self.mesh_type = path, mesh_path, mesh_type
        self._type = 'dir'
        self.load_lms_counter, self.load_bg, self.load_uv = load_lms_counter, load_bg, load_uv
        self.load_coeff = fvcoeffs_path is not None
        self.label_file = label_file
        sibling = lambda folder, wanted: path.replace('images512x512', folder) if wanted else None
        self._condImg_path = sibling('lms_counter512x512_newF', load_lms_counter)
        self._bg_path = sibling('fgmasks512x512', load_bg)
        self._uv_path = sibling('uvRender256x256', load_uv)
        self._coeff_path = fvcoeffs_path
        if load_uv:
            self.uvmask = np.asarray(PIL.Image.open(UV_FACE_MASK).convert('L')).astype(np.float32) / 255
        cams_json = os.path.join(path, 'dataset_realcam.json')
        with open(cams_json) as f:
            self._image_fnames = list(dict(json.load(f)['labels']).keys())
        self._uv_fnames = [n.split('.')[0] + '.npy' for n in self._image_fnames]
        if not self._image_fnames:
            raise IOError('No image files found in the specified path')
        self._raw_cams = self._load_raw_label(cams_json, 'labels')
        super().__init__(name=os.path.splitext(os.path.basename(path))[0],
                         raw_shape=[len(self._image_fnames), 3, resolution, resolution], **super_kwargs)

    def _load_raw_label(self, json_path, sub_key=None):
        with open(json_path, 'rb') as f:
            labels = json.load(f)
        labels = dict(labels[sub_key] if sub_key is not None else labels)
        return np.array([labels[n.replace('\\', '/')] for n in self._image_fnames]).astype(np.float32)

    def _load_raw_labels(self):
        return self._load_raw_label(os.path.join(self._path, self.label_file), 'labels')

    def _load_raw_image_core(self, fname, path=None, resolution=None):
        return _read_image_chw(os.path.join(path or self._path, fname), resolution)

    def _uv_image(self, fname, npy_name):
        """[7,256,256]: UV-space ground-truth texture (3, in [-1,1]) + projected-vertex image (4, last channel x face mask)."""
        pverts = np.load(os.path.join(self._uv_path, npy_name)).astype(np.float32)
        pverts[..., -1] *= self.uvmask
        with open(os.path.join(self._uv_path, fname.split('.')[0] + '_uvgttex.png'), 'rb') as f:
            tex = np.array(PIL.Image.open(f)).astype(np.float32) / 127.5 - 1
        return np.concatenate([tex, pverts], axis=-1).transpose(2, 0, 1)

    def _load_raw_image(self, raw_idx, resolution=None):
        fname = self._image_fnames[raw_idx]
        image = self._load_raw_image_core(fname, resolution=resolution)
        if self.load_lms_counter:
            image = np.concatenate([image, self._load_raw_image_core(fname, self._condImg_path, resolution)], axis=0)
        if self.load_bg:
            image = np.concatenate([image, _binarise_mask_image(os.path.join(self._bg_path, fname), resolution)], axis=0)
        if self.load_uv:
            image = {'image': image, 'uv': self._uv_image(fname, fname.replace('png', 'npy'))}     # (:315)
        return image

    def get_vert(self, raw_idx):
        fname = self._uv_fnames[raw_idx]
        uvc = np.load(os.path.join(self._mesh_path, fname))[..., :3]
        uvc[..., -1] = (uvc[..., -1] >= 0.5).astype(uvc.dtype)
        out = {'uvcoords_image': uvc.copy(), 'mouths_mask': np.asarray([0, 0, 1, 1], dtype=np.int32)}
        if self.load_coeff:
            out['coeff'] = np.load(os.path.join(self._coeff_path, fname)).astype(np.float32)
        return out

    def get_label(self, idx):
        raw = self._raw_idx[idx]
        return np.concatenate([self._get_raw_labels()[raw], self._raw_cams[raw]], axis=-1)

    def get_image(self, idx, resolution=None):
        return self._load_raw_image(self._raw_idx[idx], resolution=resolution)

    def get_CondImg(self, idx, resolution=None):
        assert self._condImg_path is not None
        return self._load_raw_image_core(self._image_fnames[self._raw_idx[idx]], self._condImg_path, resolution)

    def get_bgImg(self, idx, resolution=None):
        assert self._bg_path is not None
        return _binarise_mask_image(os.path.join(self._bg_path, self._image_fnames[self._raw_idx[idx]]), resolution)

    def get_uvImg(self, idx):
        assert self._uv_path is not None
        fname = self._image_fnames[idx]
        return self._uv_image(fname, fname.replace('.png', '.npy'))                                  # (:363)
