"""Loss API with the reference's names (src/utils/loss_functions.py).

Two ways to use them:

* the reference's way -- ``criterion(model_output, targets, sineweight, ...)`` on a materialised
  model output: plain tensor algebra on the GPU over [B,P,3] (three values per sample; the MLP is
  the expensive part and ran in the fused kernels);
* the fused way -- ``criterion.fused(model, Z, directions, targets, sineweight)``: model, loss and
  backward in one kernel launch (what ``RENI.training_step`` uses).  Same values, same gradients.
"""
import torch
import torch.nn.functional as F


def WeightedMSE(model_output, ground_truth, sineweight):
    """sum_b mean_{p,c} w (o - t)^2  (loss_functions.py:6-13)."""
    return (((model_output - ground_truth) ** 2) * sineweight).reshape(model_output.shape[0], -1).mean(1).sum(0)


def KLD(mu, log_var, Z_dims=1):
    """loss_functions.py:16-22."""
    kld = -0.5 * ((1 + log_var - mu.pow(2) - log_var.exp()).reshape(mu.shape[0], -1)).sum(1)
    return (kld / Z_dims).sum(0)


def WeightedCosineSimilarity(model_output, ground_truth, sineweight):
    """Cosine similarity over the pixel axis per channel, times the weight of pixel 0
    (loss_functions.py:25-32; quirk documented in SURVEY.md Appendix B2)."""
    cs = F.cosine_similarity(model_output, ground_truth, dim=1, eps=1e-20)
    return (1 - (cs * sineweight[:, 0]).mean(1)).sum(0)


def CosineSimilarity(model_output, ground_truth):
    return 1 - F.cosine_similarity(model_output, ground_truth, dim=1, eps=1e-20).mean()


class RENITrainLoss(object):
    """loss_functions.py:39-45."""

    def __call__(self, inputs, targets, sineweight):
        return WeightedMSE(inputs, targets, sineweight)

    def fused(self, model, Z, directions, targets, sineweight):
        return model.fused_loss(Z, directions, targets, sineweight, "mse")[0]


class RENIVADTrainLoss(object):
    """loss_functions.py:47-58."""

    def __init__(self, beta=1, Z_dims=None):
        self.beta = beta
        self.Z_dims = Z_dims

    def __call__(self, inputs, targets, sineweight, mu, log_var):
        mse_loss = WeightedMSE(inputs, targets, sineweight)
        kld_loss = self.beta * KLD(mu, log_var, self.Z_dims)
        return mse_loss + kld_loss, mse_loss, kld_loss

    def fused(self, model, Z, directions, targets, sineweight, mu, log_var):
        mse_loss = model.fused_loss(Z, directions, targets, sineweight, "mse")[0]
        kld_loss = self.beta * KLD(mu, log_var, self.Z_dims)  # [B,ND,3] algebra on the latents only
        return mse_loss + kld_loss, mse_loss, kld_loss


class RENITestLoss(object):
    """loss_functions.py:60-71."""

    def __init__(self, alpha=1, beta=1):
        self.alpha = alpha
        self.beta = beta

    def __call__(self, inputs, targets, sineweight, Z):
        mse_loss = WeightedMSE(inputs, targets, sineweight)
        prior_loss = self.alpha * torch.pow(Z, 2).sum()
        cosine_loss = self.beta * WeightedCosineSimilarity(inputs, targets, sineweight)
        return mse_loss + prior_loss + cosine_loss, mse_loss, prior_loss, cosine_loss

    def fused(self, model, Z, directions, targets, sineweight, sparse_weight=False):
        """sparse_weight: the weight carries an inpainting mask (RENI_module.py:92-94) -- see ops.Plan.forward_loss_backward."""
        t = model.fused_loss(Z, directions, targets, sineweight, "test", self.alpha, self.beta, sparse_weight=sparse_weight)
        return t[0], t[1], t[2], t[3]


class RENITestLossInverse(object):
    """loss_functions.py:73-85 (FIT_INVERSE: plain MSE + prior + cosine similarity on the renders)."""

    def __init__(self, alpha=1, beta=1):
        self.alpha = alpha
        self.beta = beta
        self.mse = torch.nn.MSELoss(reduction="mean")

    def __call__(self, inputs, targets, Z):
        mse_loss = self.mse(inputs, targets)
        prior_loss = self.alpha * torch.pow(Z, 2).sum()
        cosine_loss = self.beta * CosineSimilarity(inputs, targets)
        return mse_loss + prior_loss + cosine_loss, mse_loss, prior_loss, cosine_loss
