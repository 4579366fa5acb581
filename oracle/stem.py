"""oracle/stem.py — CPU restatement of ResNetEncoder + calc_mean_std + compute_resnet_features' tensor math
(retrieval/clip100_resnet_style_all_shots.py:51-74,197-200).  TEST INFRASTRUCTURE ONLY.
torchvision (resnet50 IMAGENET1K_V1) is not installed here: the stem is restated with torch.nn.functional from
its published definition (conv1 7x7/2 pad 3 no bias, BatchNorm2d eval eps 1e-5, ReLU, MaxPool 3x3/2 pad 1) and checked, bit
for bit, against `transformers`' ResNetEmbeddings — the same stem, importable here (tests/test_host_logic.py).
``calc_mean_std`` follows the reference line by line (unbiased var + eps, then sqrt)."""
import torch
import torch.nn.functional as F


def stem(x, st):
    x = F.conv2d(x, st["conv1.weight"], None, stride=2, padding=3)
    x = F.batch_norm(x, st["bn1.running_mean"], st["bn1.running_var"], st["bn1.weight"], st["bn1.bias"], False, 0.0, 1e-5)
    return F.max_pool2d(F.relu(x), 3, 2, 1)


def calc_mean_std(feat, eps=1e-5):
    N, C = feat.shape[:2]
    var = feat.view(N, C, -1).var(dim=2) + eps
    return feat.view(N, C, -1).mean(dim=2), var.sqrt()


def style_vector(x, st):
    m, s = calc_mean_std(stem(x, st))
    return torch.cat([m, s], dim=1)
