from .ctr_trainer import CTRTrainer
