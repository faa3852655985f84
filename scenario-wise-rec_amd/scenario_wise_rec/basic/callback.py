"""Early stopping on validation AUC (reference: `basic/callback.py:4-33`)."""
import copy


class EarlyStopper(object):
    """Keeps the best weights seen; asks to stop after `patience` epochs without improvement."""

    def __init__(self, patience):
        self.patience = patience
        self.trial_counter = 0
        self.best_auc = 0
        self.best_weights = None

    def stop_training(self, val_auc, weights):
        if val_auc > self.best_auc:
            self.best_auc, self.trial_counter = val_auc, 0
            self.best_weights = copy.deepcopy(weights)
            return False
        if self.trial_counter + 1 < self.patience:
            self.trial_counter += 1
            return False
        return True
