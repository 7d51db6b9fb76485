"""Training loop: drop-in for the reference `Trainer.py` on the hot-path model types.

Keeps `Trainer(model, model_type, dtype, device, output_save_dir, dataloaders, batch_size,
optimizer, patience, num_epochs, loss_function, accuracy_metric, lr_scheduler=None,
start_epoch=1)` and `.train()` (reference Trainer.py:12-50, :113-129), and the behaviour of
`singe_train` (Trainer.py:663-829): per-batch order  to(device)/type -> forward -> calc_loss ->
zero_grad -> backward -> step -> poly-LR with the pre-increment `iter_num` (:719-726); model
selection on `val_score < best_val_score` (:752); early stop on `counter > patience` (:768);
files `logs.txt`, `models/{epochN,best,last_epoch}.pt`, `total.png`.

Differences, all host-side: the step body is factored into `train_step` (reused by the
data-parallel runner, umi/ddp.py) and the running loss is accumulated on the device and read
back once per epoch instead of a `.item()` sync every step (Trainer.py:727).
`Trainer(..., graph=True)` (keyword-only, or env UMI_TRAINER_GRAPH=1) replays the whole step from a captured HIP graph
(umi.graphs.GraphedStep) when the model runs on the MI355X and the optimizer is a `umi.optim` one: the first step of a batch
shape runs eagerly, the second is captured and replayed, ragged last batches run eagerly.  The poly learning-rate rule
(:722-725) then advances inside the step from a device-resident block (`optimizer.device_schedule`), because a captured
kernel argument would freeze it; `param_groups[i]['lr']` is refreshed from the device when it is printed or saved.
`multi_task_train` (Trainer.py:831-992; model types 'multi_task*' with a plain loss name): two-headed models
(`Model.UNet_multitask`, `VisionTransformerMultitask`), batches `(inputs, (label1, label2))`, both outputs through
`F.relu` (:883-884), loss = loss1 + loss2 (:885-890), model selection on the validation LOSS (:926).
The remaining epoch loops of the reference (uncertainty / ratio weighted multi-task, CLTR, Topo losses) are out of scope
and raise.
"""
import copy
import os
import time

import torch
import torch.nn.functional as F
from tqdm import tqdm

from loss import calc_loss

_SINGLE = ('single', 'TransUnet', 'regression', 'regression_t', 'attention')
_TOPO = ('TopoCount', 'TopoCount2', 'TopoLoss', 'TopoLoss2', 'MyTopoLoss1', 'MyTopoLoss2', 'MyTopoLossGraph',
         'MyTopoLossVR')
_MULTI = ('multi_task', 'multi_task_reg', 'multi_task_regTU')
_OTHER = ('CLTR',)


class Trainer():
    def __init__(self, model, model_type, dtype, device, output_save_dir, dataloaders, batch_size, optimizer,
                 patience, num_epochs, loss_function, accuracy_metric, lr_scheduler=None, start_epoch=1, *, graph=None):
        self.model = model
        self.model_type = model_type
        self.dtype = dtype
        self.device = device
        self.output_save_dir = output_save_dir
        self.dataloader = dataloaders
        self.batch_size = batch_size
        self.optimizer = optimizer
        self.patience = patience
        self.num_epochs = num_epochs
        self.loss_function = loss_function
        self.accuracy_metric = accuracy_metric
        self.lr_scheduler = lr_scheduler
        self.start_epoch = start_epoch

        self.phases = ["train", "val"]
        self.iter_num = 0
        self.base_lr = self.optimizer.param_groups[-1]['lr']
        self.max_iterations = self.num_epochs * len(self.dataloader['train'])
        self.best_loss = 1e15
        self.best_val_score = 0 if accuracy_metric in ('dice_score', 'dice_score_mc') else 1e15
        self.best_model = []
        self.early_stop_counter = 0
        self.train_loss_list, self.val_loss_list, self.val_score_list = [], [], []
        self.train_loss_list_1, self.train_loss_list_2, self.val_loss_list_1, self.val_loss_list_2 = [], [], [], []
        self.grad_sync = None         # optional callable run between backward and optimizer.step (DDP)
        self.graph = (os.environ.get("UMI_TRAINER_GRAPH") == "1") if graph is None else bool(graph)
        self._graphs, self._seen_shapes, self._dev_sched, self._side = {}, set(), False, None

        self.save_dir_model = os.path.join(self.output_save_dir, 'models/')
        os.makedirs(self.save_dir_model, exist_ok=True)

    # ------------------------------------------------------------------------------------
    def train(self):
        if self.model_type in _SINGLE:
            if self.loss_function in _TOPO:
                raise NotImplementedError("Topo-loss warm-up loop (reference singe_train_wup) is out of scope")
            return self.singe_train()
        if self.model_type in _MULTI:
            if self.loss_function in ('multi_task_loss', 'multi_task_loss_ratio'):
                raise NotImplementedError("uncertainty- / ratio-weighted multi-task loops (reference multi_task_uc_train, "
                                          "multi_task_trainRatio) are out of scope")
            return self.multi_task_train()
        if self.model_type in _OTHER:
            raise NotImplementedError(f'model_type "{self.model_type}" (CLTR loop) is out of scope')
        raise ValueError('Invalid model_type "%s"' % self.model_type)

    # ------------------------------------------------------------------------------------
    def _to_device(self, inputs, labels):
        if isinstance(labels, (list, tuple)):                    # multi-task batches: labels = (label1, label2)
            return inputs.to(self.device).type(self.dtype), tuple(l.to(self.device).type(self.dtype) for l in labels)
        return inputs.to(self.device).type(self.dtype), labels.to(self.device).type(self.dtype)

    def _forward_loss(self, inputs, labels):
        if self.model_type in _MULTI:                            # reference Trainer.py:882-890
            outs = tuple(F.relu(o) for o in self.model(inputs))
            self._task_losses = [calc_loss(o, l, loss_type=self.loss_function) for o, l in zip(outs, labels)]
            return outs, self._task_losses[0] + self._task_losses[1]
        out = self.model(inputs)
        if self.model_type in ('regression', 'regression_t'):
            out = F.relu(out)
        return out, calc_loss(out, labels, loss_type=self.loss_function)

    def _step_body(self, inputs, labels):
        with torch.set_grad_enabled(True):
            _, loss = self._forward_loss(inputs, labels)
            self.optimizer.zero_grad()
            loss.backward()
            if self.grad_sync is not None:
                self.grad_sync()
            self.optimizer.step()
        return loss

    def _graph_capable(self, inputs):
        return (self.graph and self.grad_sync is None and inputs.is_cuda and hasattr(self.optimizer, "device_schedule"))

    def train_step(self, inputs, labels):
        """One optimisation step (reference Trainer.py:700-726).  Returns the detached loss tensor."""
        inputs, labels = self._to_device(inputs, labels)
        if self._graph_capable(inputs):
            return self._graphed_train_step(inputs, labels)
        loss = self._step_body(inputs, labels)
        if self.lr_scheduler:
            lr_ = self.base_lr * (1.0 - self.iter_num / self.max_iterations) ** 0.9
            for group in self.optimizer.param_groups:
                group['lr'] = lr_
        self.iter_num += 1
        return loss.detach()

    def _graphed_train_step(self, inputs, labels):
        from umi.graphs import GraphedStep
        if not self._dev_sched:
            poly = dict(base_lr=self.base_lr, max_iterations=self.max_iterations, power=0.9, iter_num=self.iter_num)
            self.optimizer.device_schedule(poly=poly if self.lr_scheduler else None)
            self._dev_sched, self._side = True, torch.cuda.Stream()
        flat = [inputs] + (list(labels) if isinstance(labels, (list, tuple)) else [labels])
        multi = isinstance(labels, (list, tuple))
        key = tuple(tuple(t.shape) for t in flat)

        def body(x, *ys):
            loss = self._step_body(x, tuple(ys) if multi else ys[0])
            return (loss.detach(),) + tuple(l.detach() for l in (self._task_losses if multi else ()))

        gs = self._graphs.get(key)
        if gs is None and key in self._seen_shapes:
            gs = self._graphs[key] = GraphedStep(body, flat, warmup=0, optimizers=[self.optimizer])   # 2nd step of a shape: capture, replay
        if gs is not None:
            outs = gs(*flat)
        else:
            # first step of a batch shape (allocations, weight-pack caches) or a ragged batch: eager, on the side stream; its
            # outputs are detached, so no autograd graph of it survives into a later capture (umi/graphs.py)
            self._seen_shapes.add(key)
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                outs = body(*flat)
            torch.cuda.current_stream().wait_stream(self._side)
        if multi:
            self._task_losses = [o.clone() for o in outs[1:]]
        self.iter_num += 1
        return outs[0].clone()            # the static output buffer is overwritten by the next replay

    def eval_step(self, inputs, labels):
        inputs, labels = self._to_device(inputs, labels)
        with torch.no_grad():
            out, loss = self._forward_loss(inputs, labels)
            if self.model_type in _MULTI:                        # the reference reports no validation score there (:853)
                return loss.detach(), torch.zeros((), device=loss.device)
            score = calc_loss(out, labels, loss_type=self.accuracy_metric)
        return loss.detach(), score.detach()

    # ------------------------------------------------------------------------------------
    def multi_task_train(self):
        return self.singe_train()                                # same epoch loop; the differences are flagged `multi` below

    def singe_train(self):
        multi = self.model_type in _MULTI
        os.makedirs(self.output_save_dir, exist_ok=True)
        log = open(os.path.join(self.output_save_dir, "logs.txt"), 'a')

        def say(msg, echo=True):
            if echo:
                print(msg)
            log.write(msg + "\n")

        use_cuda = torch.cuda.is_available()
        total_mem = f'{torch.cuda.get_device_properties(0).total_memory / 1E9 if use_cuda else 0:.3g}G'
        total_time = 0.0
        for epoch in range(self.start_epoch, self.num_epochs + 1):
            log.write('Epoch {}/{}\n'.format(epoch, self.num_epochs) + '-' * 10 + "\n")
            since = time.time()
            for phase in self.phases:
                train = phase == 'train'
                if train:
                    if self._dev_sched:
                        self.optimizer.sync_host()          # the LR lives on the device in graph mode
                    for group in self.optimizer.param_groups:
                        print("LR", group['lr'])
                        log.write(f"LR {group['lr']}\n")
                    since = time.time()
                self.model.train(train)

                loss_sum, score_sum, steps = None, None, 0
                task_sums = [0.0, 0.0]
                with tqdm(self.dataloader[phase], unit="batch") as bar:
                    for inputs, labels in bar:
                        bar.set_description(f"Epoch {epoch}")
                        steps += 1
                        if train:
                            loss = self.train_step(inputs, labels)
                        else:
                            loss, score = self.eval_step(inputs, labels)
                            score_sum = score if score_sum is None else score_sum + score
                        loss_sum = loss if loss_sum is None else loss_sum + loss
                        if multi:
                            task_sums = [s_ + l_.detach() for s_, l_ in zip(task_sums, self._task_losses)]
                        if steps % 50 == 0 or not use_cuda:     # avoid a device sync on every step
                            mem = f'{torch.cuda.memory_reserved() / 1E9 if use_cuda else 0:.3g}G/' + total_mem
                            bar.set_postfix(loss=float(loss_sum) / steps, memory=mem)
                epoch_loss = float(loss_sum) / steps

                if train:
                    elapsed = time.time() - since
                    say('Training Time for this epoch: {:.0f}m {:.0f}s\n'.format(elapsed // 60, elapsed % 60))
                    self.train_loss_list.append(epoch_loss)
                    if multi:
                        self.train_loss_list_1.append(float(task_sums[0]) / steps)
                        self.train_loss_list_2.append(float(task_sums[1]) / steps)
                    say("Train loss on epoch %i: %f" % (epoch, epoch_loss))
                    total_time += elapsed
                    self.meanTimePerEpoch = total_time / epoch
                    say('Curent mean training time per epoch: {:.0f}m {:.0f}s\n'.format(
                        self.meanTimePerEpoch // 60, self.meanTimePerEpoch % 60))
                    torch.save(self.model.state_dict(), os.path.join(self.save_dir_model, 'last_epoch.pt'))
                    continue

                val_score = float(score_sum) / steps
                if multi:
                    self.val_loss_list_1.append(float(task_sums[0]) / steps)
                    self.val_loss_list_2.append(float(task_sums[1]) / steps)
                self.val_loss_list.append(epoch_loss)
                self.val_score_list.append(val_score)
                say("Val loss on epoch %i: %f" % (epoch, epoch_loss))
                say("Val score on epoch %i: %f" % (epoch, val_score))
                selector = epoch_loss if multi else val_score   # multi-task: best model by validation loss (:926)
                if selector < self.best_val_score:
                    self.early_stop_counter = 0
                    self.best_val_score = selector
                    self.best_loss = epoch_loss
                    say("saving best model")
                    self.best_model = copy.deepcopy(self.model.state_dict())
                    torch.save(self.best_model, os.path.join(self.save_dir_model, 'epoch{}.pt'.format(epoch)))
                    torch.save(self.best_model, os.path.join(self.save_dir_model, 'best.pt'))
                else:
                    self.early_stop_counter += 1
                if self.early_stop_counter > self.patience:
                    say("Early stopping")
                    return self._finish(log, say)

            elapsed = time.time() - since
            say('{:.0f}m {:.0f}s\n'.format(elapsed // 60, elapsed % 60))
        return self._finish(log, say)

    def _finish(self, log, say):
        say('Best val loss: {:4f}'.format(self.best_loss))
        say('Best val score: {:4f}'.format(self.best_val_score))
        log.close()
        self.plot_loss_functions('total')
        if self.best_model:
            self.model.load_state_dict(self.best_model)     # load best model weights
        return self.model

    def plot_loss_functions(self, name):
        """Train/val loss curves -> <output_save_dir>/<name>.png (reference Trainer.py:52-111)."""
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(figsize=(8, 5))
        ep = range(1, len(self.train_loss_list) + 1)
        ax.plot(ep, self.train_loss_list, label='train loss')
        if self.val_loss_list:
            ax.plot(range(1, len(self.val_loss_list) + 1), self.val_loss_list, label='validation loss')
        ax.set_xlabel('epoch')
        ax.set_ylabel('loss')
        ax.legend()
        fig.savefig(os.path.join(self.output_save_dir, name + '.png'))
        plt.close(fig)
