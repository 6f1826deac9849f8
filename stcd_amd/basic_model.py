"""``CDEvaluator`` -- inference wrapper with the interface of /root/reference/models/basic_model.py:9-74
(build the network, load ``best_ckpt.pt``, forward, ``argmax * 255``, write PNGs).  The reference's ``misc.imutils``
is missing from its repository; PNGs are written with Pillow here."""
from __future__ import annotations

import os

import numpy as np
import torch

from .networks import define_G


def save_image(image_numpy, image_path):
    from PIL import Image

    Image.fromarray(np.asarray(image_numpy).astype(np.uint8)).save(image_path)


class CDEvaluator:
    def __init__(self, args):
        self.n_class = args.n_class
        self.net_G = define_G(args=args, gpu_ids=args.gpu_ids)
        self.device = torch.device("cuda:%s" % args.gpu_ids[0] if torch.cuda.is_available() and len(args.gpu_ids) > 0 else "cpu")
        print(self.device)
        self.checkpoint_dir = args.checkpoint_dir
        self.pred_dir = args.output_folder
        os.makedirs(self.pred_dir, exist_ok=True)

    def load_checkpoint(self, checkpoint_name="best_ckpt.pt"):
        path = os.path.join(self.checkpoint_dir, checkpoint_name)
        if not os.path.exists(path):
            raise FileNotFoundError("no such checkpoint %s" % checkpoint_name)
        checkpoint = torch.load(path, map_location=self.device, weights_only=False)
        self.net_G.load_state_dict(checkpoint["model_G_state_dict"])
        self.net_G.to(self.device)
        self.best_val_acc = checkpoint["best_val_acc"]
        self.best_epoch_id = checkpoint["best_epoch_id"]
        return self.net_G

    def _visualize_pred(self):
        return torch.argmax(self.G_pred, dim=1, keepdim=True) * 255

    def _forward_pass(self, batch):
        self.batch = batch
        img_in1 = batch["A"].to(self.device)
        img_in2 = batch["B"].to(self.device)
        self.shape_h, self.shape_w = img_in1.shape[-2], img_in1.shape[-1]
        out = self.net_G(img_in1, img_in2)
        self.G_pred = out[-1] if isinstance(out, (list, tuple)) else out    # tensor-returning models
        return self._visualize_pred()

    def eval(self):
        self.net_G.eval()

    def _save_predictions(self):
        preds = self._visualize_pred()
        for i, pred in enumerate(preds):
            file_name = os.path.join(self.pred_dir, self.batch["name"][i].replace(".jpg", ".png"))
            save_image(pred[0].cpu().numpy(), file_name)
