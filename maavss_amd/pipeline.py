"""Two-stream clip pipeline: attention-frame extraction + STFT of batch i+1 on a side HIP stream while the training step of
batch i runs on the main stream.

In the reference the ViT and the STFT run in the DATA path (`AV_Dataset.__getitem__`, av_dataset.py:321 and :335-342, called
by the DataLoader at train_avse_frames.py:122) and have no dependency on the optimizer step (:150-181) -- only the order
"extract batch i before training on batch i".  This module keeps exactly that dependency and nothing more: extraction of the
next batch is enqueued on its own stream and joins the training stream through HIP events, so the latency-bound parts of the
fusion network's step (16 sequential LSTM launches per direction, the M = batch Linear layers, the small STFT-encoder
convolutions) run next to the extractor's full-chip GEMMs instead of in front of them.  Slots are double-buffered; a slot
is reused only after the training step that read it has been enqueued and has signalled its `consumed` event.
"""
import torch


class ClipPipeline:
    def __init__(self, video_attention, stft, clip_frames, depth=2, finite_check="deferred"):
        self.va, self.stft, self.t = video_attention, stft, clip_frames
        self.depth = depth
        self.finite_check = finite_check
        self.side = torch.cuda.Stream()
        self.slots = [dict(attn=None, x=None, y=None, ready=torch.cuda.Event(), consumed=None) for _ in range(depth)]
        self.head = self.tail = 0            # next slot to submit into / next slot to hand out

    def submit(self, frames, audio, seed):
        """Enqueue the extraction of one batch: frames [B*T,3,H,W], audio [B,L] (both resident on the device).  The caller's
        stream must already hold the work that produced them (the side stream waits for it)."""
        assert self.head - self.tail < self.depth, "pipeline full: get()/release() a batch first"
        slot = self.slots[self.head % self.depth]
        f, _, h, w = frames.shape
        main = torch.cuda.current_stream()
        produced = torch.cuda.Event()
        produced.record(main)
        # the caller may drop its references right after this call: tell the caching allocator that the side stream still reads them
        frames.record_stream(self.side)
        audio.record_stream(self.side)
        with torch.cuda.stream(self.side):
            self.side.wait_event(produced)
            if slot["consumed"] is not None:
                self.side.wait_event(slot["consumed"])      # the training step that read this slot's buffers is past them
            if slot["attn"] is None or slot["attn"].shape != (f, 1, h, w):
                slot["attn"] = torch.empty(f, 1, h, w, device=frames.device, dtype=torch.float32)
            self.va.attention_frames(frames, clip_frames=self.t, out=slot["attn"], finite_check=self.finite_check)
            slot["x"], slot["y"] = self.stft(audio, seed=seed)    # replaces (frees) the tensors of two batches ago, after the wait above
            slot["ready"].record(self.side)
        self.head += 1

    def get(self):
        """(attention frames [B,1,T,H,W], x_stft, y_stft) of the oldest submitted batch; the current stream waits for them."""
        assert self.tail < self.head, "nothing submitted"
        slot = self.slots[self.tail % self.depth]
        torch.cuda.current_stream().wait_event(slot["ready"])
        f, _, h, w = slot["attn"].shape
        return slot["attn"].view(f // self.t, 1, self.t, h, w), slot["x"], slot["y"]

    def release(self):
        """Call after the work that reads the batch handed out by get() has been enqueued on the current stream."""
        slot = self.slots[self.tail % self.depth]
        slot["consumed"] = torch.cuda.Event()
        slot["consumed"].record(torch.cuda.current_stream())
        self.tail += 1

    def drain(self):
        self.side.synchronize()
        self.va.check_finite()
