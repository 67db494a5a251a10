// Package gpu binds libmzd.so (include/mzd.h) for sparkzstd: the MI355X hot path underneath
// decompression.FrameReader / FrameDecompressor.
//
// UNVERIFIED: written against Go 1.x cgo rules but never compiled -- the image this repository is built in has no
// Go toolchain (`go version`: command not found).  The verified drivers of the same ABI are the C++ planner
// (sparkzstd_amd/csrc/planner.cpp, include/sparkzstd_frame.hpp, tools/verify) and the Python mirror
// (sparkzstd_amd/decompression.py); this file follows them call for call.
//
// Two routes into the device:
//   - DecodeFrames / Context.DecodeFrames: whole frames in, headers parsed ON THE DEVICE
//     (mzd_batch_upload_frames, k_parse): no Go-side planning at all.  This is what reader_gpu.go uses.
//   - Batch: the host-planned route north_star describes (Go keeps frame / block / section header parsing and table
//     description parsing; decompression.PlanNextBlock fills a Batch, Batch.Decode makes ONE cgo call).
//
// cgo pointer rules: every C struct handed over is flat (no Go pointers inside), and slices are passed as
// unsafe.Pointer(&s[0]) for the duration of one call only; the library copies what it keeps.
package gpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../sparkzstd_amd -lmzd -Wl,-rpath,${SRCDIR}/../../../sparkzstd_amd
#include <stdlib.h>
#include "mzd.h"
*/
import "C"

import (
	"errors"
	"io"
	"runtime"
	"sync"
	"unsafe"
)

// Sentinels this package adds (everything else maps onto sparkzstd's own, see SentinelFor).
var (
	ErrNoDevice         = errors.New("mzd: no HIP device (the library has no CPU fallback)")
	ErrDevice           = errors.New("mzd: HIP runtime error")
	ErrUnsupported      = errors.New("mzd: frame is outside the device path's limits")
	ErrChecksumMismatch = errors.New("mzd: content checksum mismatch")
	ErrDstFull          = errors.New("mzd: frame output does not fit its slab / content size mismatch")
)

// Sentinels of sparkzstd the statuses map back to.  The shim cannot import them without creating an import cycle
// (decompression imports gpu), so the decompression package registers them at init time (reader_gpu.go: init()).
var Sentinels = map[int]error{}

// BuildID names the library the process runs on: the first 16 hex digits of the sha256 of the sources it was built from (mzd_build_id,
// ABI 8; ABI 9 adds a frame in chunks: FrameStream).
func BuildID() string { return C.GoString(C.mzd_build_id()) }

// StatusFor is SentinelFor's inverse for the errors a binding has to hand DOWN (mzd_block_desc.seq_status): 0 if the error is none
// of the registered sentinels.
func StatusFor(err error) uint8 {
	for code, e := range Sentinels {
		if e == err {
			return uint8(code)
		}
	}
	return 0
}

// SentinelFor turns a per-frame status of the device into the Go error the reference would have returned
// (INTEGRATION.md section 2, table "Mapping of statuses"); nil for MZD_OK.
func SentinelFor(status int32) error {
	if status == C.MZD_OK {
		return nil
	}
	if e, ok := Sentinels[int(status)]; ok {
		return e
	}
	switch status {
	case C.MZD_ERR_TRUNCATED:
		return io.ErrUnexpectedEOF
	case C.MZD_ERR_UNSUPPORTED:
		return ErrUnsupported
	case C.MZD_ERR_CHECKSUM:
		return ErrChecksumMismatch
	case C.MZD_ERR_DST_FULL:
		return ErrDstFull
	}
	return errors.New("mzd: " + C.GoString(C.mzd_strerror(C.int(status))))
}

// Context owns one mzd_ctx (one GPU).  Like the reference's FrameDecompressor it is NOT goroutine-safe
// (framedecompressor.go:26-31 shares scratch arrays the same way): one goroutine at a time, or one Context each.
type Context struct {
	c  *C.mzd_ctx
	mu sync.Mutex
}

// NewContext creates the device context; Close (or the finalizer) destroys it.
func NewContext(device int) (*Context, error) {
	var rc C.int
	var opt C.mzd_options // zero value = defaults: the library picks its kernels per batch (one large frame: its blocks side by side)
	c := C.mzd_create(C.int(device), &opt, &rc)
	if c == nil {
		if rc == C.MZD_ERR_NO_DEVICE {
			return nil, ErrNoDevice
		}
		return nil, ErrDevice
	}
	ctx := &Context{c: c}
	runtime.SetFinalizer(ctx, func(x *Context) { x.Close() })
	return ctx, nil
}

// Close releases the device context.  Safe to call twice.
func (x *Context) Close() {
	x.mu.Lock()
	defer x.mu.Unlock()
	if x.c != nil {
		C.mzd_destroy(x.c)
		x.c = nil
	}
}

var (
	defaultOnce sync.Once
	defaultCtx  *Context
	defaultErr  error
)

// Default returns the process-wide context on device 0 (created on first use).
func Default() (*Context, error) {
	defaultOnce.Do(func() { defaultCtx, defaultErr = NewContext(0) })
	return defaultCtx, defaultErr
}

func (x *Context) lastError() string { return C.GoString(C.mzd_last_error(x.c)) }

// concat lays the frames end to end in one blob (the layout mzd_batch_upload_frames wants).
func concat(frames [][]byte) (blob []byte, off, ln []uint64) {
	total := 0
	for _, f := range frames {
		total += len(f)
	}
	blob = make([]byte, 0, total+1)
	off = make([]uint64, len(frames))
	ln = make([]uint64, len(frames))
	for i, f := range frames {
		off[i], ln[i] = uint64(len(blob)), uint64(len(f))
		blob = append(blob, f...)
	}
	if len(blob) == 0 {
		blob = append(blob, 0) // &blob[0] must exist
	}
	return
}

// ResidentFrame is ONE frame decoded and left in HBM: what decompression.FrameReader holds between its Reads when the consumer
// takes a large frame piece by piece (framereader.go:51-109 copies the decoded bytes into p; here ReadAt copies them from the
// device into p, once, without a host copy of the whole frame in between -- mzd_batch_read_out).
type ResidentFrame struct {
	x   *Context
	db  *C.mzd_dbatch
	off uint64 // the frame's slab in the output blob
	Len uint64 // regenerated bytes
}

// DecodeFrameResident plans `frame` on the device, decodes it and returns it resident; Free releases the device memory.
func (x *Context) DecodeFrameResident(frame []byte) (*ResidentFrame, error) {
	if len(frame) == 0 {
		return nil, SentinelFor(1)
	}
	x.mu.Lock()
	defer x.mu.Unlock()
	off, ln := C.uint64_t(0), C.uint64_t(len(frame))
	var db *C.mzd_dbatch
	rc := C.mzd_batch_upload_frames(x.c, (*C.uint8_t)(unsafe.Pointer(&frame[0])), ln, 0, &off, &ln, 1, nil, 0, &db)
	if rc != C.MZD_OK {
		return nil, errors.New("mzd_batch_upload_frames: " + x.lastError())
	}
	var status C.int32_t
	var outLen, slab C.uint64_t
	if rc = C.mzd_batch_run(x.c, db, nil); rc == C.MZD_OK {
		rc = C.mzd_batch_download(x.c, db, nil, &status, &outLen)
	}
	if rc == C.MZD_OK {
		rc = C.mzd_batch_frame_layout(db, &slab, nil)
	}
	if rc == C.MZD_OK {
		// the frame's bytes stay in HBM until the reader has handed them out; the input copy, the sequence records and block
		// mode's planes (three times the output) do not
		rc = C.mzd_batch_trim(x.c, db)
	}
	runtime.KeepAlive(frame)
	if rc != C.MZD_OK {
		C.mzd_batch_free(x.c, db)
		return nil, errors.New("mzd_batch_run: " + x.lastError())
	}
	if e := SentinelFor(int32(status)); e != nil {
		C.mzd_batch_free(x.c, db)
		return nil, e
	}
	return &ResidentFrame{x: x, db: db, off: uint64(slab), Len: uint64(outLen)}, nil
}

// ReadAt copies bytes [pos, pos+len(p)) of the frame into p (clipped to the frame's end) and returns how many.
func (r *ResidentFrame) ReadAt(p []byte, pos uint64) (int, error) {
	if r.db == nil || pos >= r.Len || len(p) == 0 {
		return 0, nil
	}
	n := uint64(len(p))
	if n > r.Len-pos {
		n = r.Len - pos
	}
	r.x.mu.Lock()
	defer r.x.mu.Unlock()
	if rc := C.mzd_batch_read_out(r.x.c, r.db, C.uint64_t(r.off+pos), (*C.uint8_t)(unsafe.Pointer(&p[0])), C.uint64_t(n)); rc != C.MZD_OK {
		return 0, errors.New("mzd_batch_read_out: " + r.x.lastError())
	}
	return int(n), nil
}

// Free releases the frame's device memory (idempotent).
func (r *ResidentFrame) Free() {
	if r.db != nil {
		r.x.mu.Lock()
		C.mzd_batch_free(r.x.c, r.db)
		r.x.mu.Unlock()
		r.db = nil
	}
}

// FrameStream is ONE frame going through the device in chunks of whole blocks (mzd_fstream_*, ABI 9): the device keeps the
// frame's window and its offset history between two chunks and nothing else -- FrameDecompressor.DecodeNextBlock + Ringbuffer
// (framedecompressor.go:198-303, ringbuffer.go:36-49) for one frame.  A frame may be larger than the device's memory, its source
// may arrive piecewise, and its first bytes are out before its last ones are in.
type FrameStream struct {
	x    *Context
	fs   *C.mzd_fstream
	Done bool // the frame's last chunk has been handed out
}

// NewFrameStream: chunkOut = the bytes a chunk may regenerate (0: 64 MiB).
func (x *Context) NewFrameStream(chunkOut uint64) (*FrameStream, error) {
	x.mu.Lock()
	defer x.mu.Unlock()
	var fs *C.mzd_fstream
	if rc := C.mzd_fstream_open(x.c, C.uint64_t(chunkOut), &fs); rc != C.MZD_OK {
		return nil, errors.New("mzd_fstream_open: " + x.lastError())
	}
	return &FrameStream{x: x, fs: fs}, nil
}

// Next is one step of the stream's two-stage pipeline (include/mzd.h, mzd_fstream_next): the next chunk of whole blocks in src
// goes to the device, the bytes that come back in dst (at least 128 KiB, the same size every call) are those of the chunk the
// call before took.  consumed == 0 && produced == 0 && !Done: src holds no whole block yet.
func (s *FrameStream) Next(src, dst []byte) (consumed, produced int, err error) {
	s.x.mu.Lock()
	defer s.x.mu.Unlock()
	var used, made C.uint64_t
	var done C.int
	var sp *C.uint8_t
	if len(src) > 0 {
		sp = (*C.uint8_t)(unsafe.Pointer(&src[0]))
	}
	rc := C.mzd_fstream_next(s.fs, sp, C.uint64_t(len(src)), (*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint64_t(len(dst)), &used, &made, &done)
	runtime.KeepAlive(src)
	runtime.KeepAlive(dst)
	if rc != C.MZD_OK {
		if e := SentinelFor(int32(rc)); e != nil {
			return int(used), 0, e
		}
		return int(used), 0, errors.New("mzd_fstream_next: " + s.x.lastError())
	}
	s.Done = done != 0
	return int(used), int(made), nil
}

// Close releases the stream's device memory (idempotent).
func (s *FrameStream) Close() {
	if s.fs != nil {
		s.x.mu.Lock()
		C.mzd_fstream_close(s.fs)
		s.x.mu.Unlock()
		s.fs = nil
	}
}

// DecodeFrames decodes many independent zstd frames in ONE device batch (the entry where the device pays off;
// the reference decodes one frame per FrameReader, cmd/sparkzstd/main.go:59,126).  out[i] is nil where errs[i] != nil.
func (x *Context) DecodeFrames(frames [][]byte) (out [][]byte, errs []error, err error) {
	n := len(frames)
	out, errs = make([][]byte, n), make([]error, n)
	if n == 0 {
		return out, errs, nil
	}
	x.mu.Lock()
	defer x.mu.Unlock()
	if x.c == nil {
		return nil, nil, ErrNoDevice
	}
	blob, off, ln := concat(frames)
	var db *C.mzd_dbatch
	rc := C.mzd_batch_upload_frames(x.c, (*C.uint8_t)(unsafe.Pointer(&blob[0])), C.uint64_t(len(blob)), 0,
		(*C.uint64_t)(unsafe.Pointer(&off[0])), (*C.uint64_t)(unsafe.Pointer(&ln[0])), C.uint32_t(n), nil, 0, &db)
	if rc != C.MZD_OK {
		return nil, nil, errors.New("mzd_batch_upload_frames: " + C.GoString(C.mzd_strerror(rc)) + ": " + x.lastError())
	}
	defer C.mzd_batch_free(x.c, db)
	if rc = C.mzd_batch_run(x.c, db, nil); rc != C.MZD_OK {
		return nil, nil, errors.New("mzd_batch_run: " + x.lastError())
	}
	slab := make([]byte, uint64(C.mzd_batch_out_size(db))+1)
	status := make([]int32, n)
	outLen := make([]uint64, n)
	slabOff := make([]uint64, n)
	rc = C.mzd_batch_download(x.c, db, (*C.uint8_t)(unsafe.Pointer(&slab[0])),
		(*C.int32_t)(unsafe.Pointer(&status[0])), (*C.uint64_t)(unsafe.Pointer(&outLen[0])))
	if rc != C.MZD_OK {
		return nil, nil, errors.New("mzd_batch_download: " + x.lastError())
	}
	C.mzd_batch_frame_layout(db, (*C.uint64_t)(unsafe.Pointer(&slabOff[0])), nil)
	for i := 0; i < n; i++ {
		if errs[i] = SentinelFor(status[i]); errs[i] == nil {
			out[i] = slab[slabOff[i] : slabOff[i]+outLen[i] : slabOff[i]+outLen[i]]
		}
	}
	runtime.KeepAlive(blob)
	return out, errs, nil
}

// DecodeFrames on the default context.
func DecodeFrames(frames [][]byte) ([][]byte, []error) {
	x, err := Default()
	if err != nil {
		errs := make([]error, len(frames))
		for i := range errs {
			errs[i] = err
		}
		return make([][]byte, len(frames)), errs
	}
	out, errs, err := x.DecodeFrames(frames)
	if err != nil {
		errs = make([]error, len(frames))
		for i := range errs {
			errs[i] = err
		}
		return make([][]byte, len(frames)), errs
	}
	return out, errs
}

// ---------------------------------------------------------------------------------------------------------
// One batch over several GPUs of a node.  Frames share nothing (tables, offset history and window are per frame:
// framedecompressor.go:42-52), so device r takes a contiguous range of them -- equal counts when the frames cost the
// same, equal C + D otherwise -- on its own Context and goroutine; results are stitched in frame order.  No collective.

var (
	poolMu sync.Mutex
	pool   = map[[2]int]*Context{} // (device, k-th context on it) -> context
)

func poolContext(device, k int) (*Context, error) {
	poolMu.Lock()
	defer poolMu.Unlock()
	if x, ok := pool[[2]int{device, k}]; ok {
		return x, nil
	}
	x, err := NewContext(device)
	if err == nil {
		pool[[2]int{device, k}] = x
	}
	return x, err
}

// DeclaredFrameCost is C + D of one frame from its header alone (frame.go:23-61); a frame that declares no content size
// counts with its window, anything that is not a frame with its length.
func DeclaredFrameCost(f []byte) uint64 {
	n := uint64(len(f))
	if n < 6 || f[0] != 0x28 || f[1] != 0xB5 || f[2] != 0x2F || f[3] != 0xFD {
		return n
	}
	fhd := f[4]
	fcsFlag, single, did := fhd>>6, (fhd>>5)&1, fhd&3
	pos := 5
	window := uint64(128 * 1024)
	if single == 0 {
		wd := f[pos]
		base := uint64(1) << (10 + (wd >> 3))
		window = base + (base>>3)*uint64(wd&7)
		pos++
	}
	pos += []int{0, 1, 2, 4}[did]
	fcsBytes := []int{0, 2, 4, 8}[fcsFlag]
	if fcsFlag == 0 && single == 1 {
		fcsBytes = 1
	}
	if fcsBytes == 0 || pos+fcsBytes > len(f) {
		return n + window
	}
	var d uint64
	for i := 0; i < fcsBytes; i++ {
		d |= uint64(f[pos+i]) << (8 * uint(i))
	}
	if fcsBytes == 2 {
		d += 256
	}
	return n + d
}

// ShardFrames returns the frame range [lo, hi) of each of `world` devices.
func ShardFrames(frames [][]byte, world int) [][2]int {
	n := len(frames)
	r := make([][2]int, world)
	cost := make([]uint64, n)
	same := true
	total := 0.0
	for i, f := range frames {
		cost[i] = DeclaredFrameCost(f)
		same = same && cost[i] == cost[0]
		total += float64(cost[i])
	}
	if same {
		base, extra, lo := n/world, n%world, 0
		for k := 0; k < world; k++ {
			hi := lo + base
			if k < extra {
				hi++
			}
			r[k] = [2]int{lo, hi}
			lo = hi
		}
		return r
	}
	acc, target, lo := 0.0, 0.0, 0
	for k := 0; k < world; k++ {
		target += total / float64(world)
		hi := lo
		for hi < n && (acc+float64(cost[hi]) <= target || hi == lo) && n-hi > world-k-1 {
			acc += float64(cost[hi])
			hi++
		}
		if k == world-1 {
			hi = n
		}
		r[k] = [2]int{lo, hi}
		lo = hi
	}
	return r
}

// DecodeFramesOn decodes ONE batch of independent frames on the listed devices (a device listed twice gets two
// contexts).  out[i] is nil where errs[i] != nil; a device that fails as a whole fails the frames of its range.
func DecodeFramesOn(devices []int, frames [][]byte) ([][]byte, []error) {
	if len(devices) == 0 {
		return DecodeFrames(frames)
	}
	out, errs := make([][]byte, len(frames)), make([]error, len(frames))
	ranges := ShardFrames(frames, len(devices))
	seen := map[int]int{}
	var wg sync.WaitGroup
	for k, d := range devices {
		lo, hi := ranges[k][0], ranges[k][1]
		x, err := poolContext(d, seen[d])
		seen[d]++
		if hi <= lo {
			continue
		}
		if err != nil {
			for i := lo; i < hi; i++ {
				errs[i] = err
			}
			continue
		}
		wg.Add(1)
		go func(x *Context, lo, hi int) {
			defer wg.Done()
			o, e, err := x.DecodeFrames(frames[lo:hi])
			for i := lo; i < hi; i++ {
				if err != nil {
					errs[i] = err
				} else {
					out[i], errs[i] = o[i-lo], e[i-lo]
				}
			}
		}(x, lo, hi)
	}
	wg.Wait()
	return out, errs
}

// ---------------------------------------------------------------------------------------------------------
// The host-planned route: Go parses headers and table descriptions (north_star), the device runs the three loops.

// Table kinds of mzd_fse_table_desc.kind.
const (
	KindLL = 0
	KindOF = 1
	KindML = 2
)

// Batch accumulates descriptors for many frames.  All slices are flat and pointer-free.
type Batch struct {
	In         []byte // the block payloads the descriptors point into (append-only)
	frames     []C.mzd_frame_desc
	blocks     []C.mzd_block_desc
	fseTables  []C.mzd_fse_table_desc
	fseEntries []C.mzd_fse_entry
	hufTables  []C.mzd_huf_table_desc
	hufEntries []C.mzd_huf_entry
	outSize    uint64
}

// Append copies payload bytes into the batch's input blob and returns their offset.
func (b *Batch) Append(p []byte) uint64 {
	at := uint64(len(b.In))
	b.In = append(b.In, p...)
	return at
}

// BeginFrame opens a frame (frame.go:23-61 has been parsed by the caller): bound = Frame_Content_Size when the
// header carries one, else blocks x 128 KiB once known (FinishFrame patches it).  Returns the frame index.
func (b *Batch) BeginFrame(windowSize uint64, contentSize uint64, hasContentSize bool) int {
	f := C.mzd_frame_desc{first_block: C.uint32_t(len(b.blocks)), window_size: C.uint64_t(windowSize)}
	if hasContentSize {
		f.content_size = C.uint64_t(contentSize)
	} else {
		f.content_size = C.MZD_UNKNOWN_SIZE
	}
	b.frames = append(b.frames, f)
	return len(b.frames) - 1
}

// FinishFrame closes the frame opened last: counts its blocks and reserves its output slab (256-byte aligned).
func (b *Batch) FinishFrame() {
	f := &b.frames[len(b.frames)-1]
	f.n_blocks = C.uint32_t(len(b.blocks)) - f.first_block
	capacity := uint64(f.n_blocks) * 128 * 1024
	if f.content_size != C.MZD_UNKNOWN_SIZE {
		capacity = uint64(f.content_size)
	}
	f.out_offset = C.uint64_t(b.outSize)
	f.out_capacity = C.uint64_t(capacity)
	b.outSize += (capacity + 255) &^ 255
}

// AddRawBlock / AddRLEBlock: framedecompressor.go:211-215 / :229-241.
func (b *Batch) AddRawBlock(payload []byte) {
	b.blocks = append(b.blocks, C.mzd_block_desc{_type: C.MZD_BLOCK_RAW, size: C.uint32_t(len(payload)), src_off: C.uint64_t(b.Append(payload))})
}
func (b *Batch) AddRLEBlock(value byte, size uint32) {
	b.blocks = append(b.blocks, C.mzd_block_desc{_type: C.MZD_BLOCK_RLE, size: C.uint32_t(size), src_off: C.uint64_t(b.Append([]byte{value}))})
}

// CompressedBlock is what decompression.PlanNextBlock has parsed of one Compressed block; AddCompressedBlock turns
// it into a mzd_block_desc.  Table fields are indices returned by the Add*Table methods (MZD_NO_TABLE = 0xFFFFFFFF
// where a section has none).
type CompressedBlock struct {
	BlockSize                 uint32
	LitType                   int       // MZD_LIT_RAW / _RLE / _HUF (Compressed and Treeless both arrive as HUF)
	LitStreams                int       // 1 or 4
	LitRegen                  uint32    // literals.go:30-41 RegeneratedSize
	LitPayload                []byte    // raw literals / the RLE byte / the 1 or 4 Huffman streams back to back
	LitStreamSize             [4]uint32 // literals.go:46-62 (jump table + computed 4th)
	HufTable                  uint32
	NSeq                      uint32 // sequences.go:378-400
	SeqStatus                 uint8  // NSeq == 0 in its two-byte form: what DecodeSequences made of the section (0, 8, 11); mzd_block_desc.seq_status
	SeqPayload                []byte // the sequence bitstream (ss.Data)
	LLTable, OFTable, MLTable uint32
}

func (b *Batch) AddCompressedBlock(cb *CompressedBlock) {
	d := C.mzd_block_desc{_type: C.MZD_BLOCK_COMPRESSED, lit_type: C.uint8_t(cb.LitType), lit_streams: C.uint8_t(cb.LitStreams),
		size: C.uint32_t(cb.BlockSize), lit_regen: C.uint32_t(cb.LitRegen), lit_off: C.uint64_t(b.Append(cb.LitPayload)),
		huf_table: C.uint32_t(cb.HufTable), n_seq: C.uint32_t(cb.NSeq), seq_status: C.uint8_t(cb.SeqStatus),
		ll_table: C.uint32_t(cb.LLTable), of_table: C.uint32_t(cb.OFTable), ml_table: C.uint32_t(cb.MLTable)}
	for i := 0; i < 4; i++ {
		d.lit_stream_size[i] = C.uint32_t(cb.LitStreamSize[i])
	}
	if cb.NSeq > 0 {
		d.seq_off, d.seq_size = C.uint64_t(b.Append(cb.SeqPayload)), C.uint32_t(len(cb.SeqPayload))
	}
	b.blocks = append(b.blocks, d)
}

// AddFSECounts ships a Compressed-mode table as its normalised counts (what
// FSETable.ReadTabledescriptionFromBitstream, fse.go:28-130, has just parsed) and leaves BuildDecodingTable
// (fse.go:136-230) to the device (k_fse_build): two int16 per cell slot, -1 = "less than one".
func (b *Batch) AddFSECounts(accLog int, counts []int16, kind int) uint32 {
	d := C.mzd_fse_table_desc{entries_off: C.uint32_t(len(b.fseEntries)), acc_log: C.uint8_t(accLog), kind: C.uint8_t(kind),
		build: C.uint16_t(C.MZD_FSE_FROM_COUNTS | len(counts))}
	for i := 0; i < len(counts); i += 2 {
		e := C.mzd_fse_entry{baseline: C.uint16_t(uint16(counts[i]))}
		if i+1 < len(counts) {
			e.nbits, e.symbol = C.uint8_t(uint16(counts[i+1])&0xFF), C.uint8_t(uint16(counts[i+1])>>8)
		}
		b.fseEntries = append(b.fseEntries, e)
	}
	b.fseTables = append(b.fseTables, d)
	return uint32(len(b.fseTables) - 1)
}

// AddFSECells copies a BUILT table (fse.go:17-24 FSETable.DecodingTable): baseline, number of bits and the
// UNtranslated symbol of every cell (keep it next to Symbol in BuildDecodingTable, fse.go:192-228: the device
// applies predefined.go:5-20,36-50 itself).
func (b *Batch) AddFSECells(accLog int, baseline []uint16, nbits []uint8, rawSymbol []uint8, kind int) uint32 {
	d := C.mzd_fse_table_desc{entries_off: C.uint32_t(len(b.fseEntries)), acc_log: C.uint8_t(accLog), kind: C.uint8_t(kind)}
	for i := range baseline {
		b.fseEntries = append(b.fseEntries, C.mzd_fse_entry{baseline: C.uint16_t(baseline[i]), nbits: C.uint8_t(nbits[i]), symbol: C.uint8_t(rawSymbol[i])})
	}
	b.fseTables = append(b.fseTables, d)
	return uint32(len(b.fseTables) - 1)
}

// AddRLETable is sequences.go:27-62 RepeatingDecodingTable: a one-cell table with acc_log 0.
func (b *Batch) AddRLETable(code byte, kind int) uint32 {
	d := C.mzd_fse_table_desc{entries_off: C.uint32_t(len(b.fseEntries)), acc_log: 0, kind: C.uint8_t(kind)}
	b.fseEntries = append(b.fseEntries, C.mzd_fse_entry{symbol: C.uint8_t(code)})
	b.fseTables = append(b.fseTables, d)
	return uint32(len(b.fseTables) - 1)
}

// AddHuffmanWeights ships the decoded weights (HuffmanTreeDesc.Weights, huffman.go:40-107, without the inferred
// last one) and leaves Build (huffman.go:112-190) to the device (k_huf_build); maxBits as in :125.
func (b *Batch) AddHuffmanWeights(weights []byte, maxBits int) uint32 {
	if len(b.hufEntries)%2 == 1 {
		b.hufEntries = append(b.hufEntries, C.mzd_huf_entry{})
	}
	d := C.mzd_huf_table_desc{entries_off: C.uint32_t(len(b.hufEntries)),
		max_bits: C.uint32_t(C.MZD_HUF_FROM_WEIGHTS | uint32(len(weights))<<8 | uint32(maxBits))}
	for i := 0; i < len(weights); i += 2 {
		e := C.mzd_huf_entry{symbol: C.uint8_t(weights[i])}
		if i+1 < len(weights) {
			e.nbits = C.uint8_t(weights[i+1])
		}
		b.hufEntries = append(b.hufEntries, e)
	}
	b.hufTables = append(b.hufTables, d)
	return uint32(len(b.hufTables) - 1)
}

// AddHuffmanCells copies a BUILT structure.HuffmanDecodingTable (huffman.go:30-37).
func (b *Batch) AddHuffmanCells(symbols []int, numberOfBits []int, maxBits int) uint32 {
	if len(b.hufEntries)%2 == 1 {
		b.hufEntries = append(b.hufEntries, C.mzd_huf_entry{})
	}
	d := C.mzd_huf_table_desc{entries_off: C.uint32_t(len(b.hufEntries)), max_bits: C.uint32_t(maxBits)}
	for i := range symbols {
		b.hufEntries = append(b.hufEntries, C.mzd_huf_entry{symbol: C.uint8_t(symbols[i]), nbits: C.uint8_t(numberOfBits[i])})
	}
	b.hufTables = append(b.hufTables, d)
	return uint32(len(b.hufTables) - 1)
}

func ptrOrNil[T any](s []T) *T {
	if len(s) == 0 {
		return nil
	}
	return &s[0]
}

// Decode runs the whole batch on the device: one synchronous cgo call (mzd_decode_batch = upload + kernels +
// download).  Returns the output blob; frame i is out[OutOffset(i) : +outLen[i]] when status[i] == 0.
func (b *Batch) Decode(x *Context) (out []byte, status []int32, outLen []uint64, err error) {
	n := len(b.frames)
	status, outLen = make([]int32, n+1), make([]uint64, n+1)
	out = make([]byte, b.outSize+256)
	x.mu.Lock()
	defer x.mu.Unlock()
	if x.c == nil {
		return nil, nil, nil, ErrNoDevice
	}
	in := b.In
	if len(in) == 0 {
		in = []byte{0}
	}
	cb := C.mzd_batch{abi_version: C.MZD_ABI_VERSION,
		in: (*C.uint8_t)(unsafe.Pointer(&in[0])), in_size: C.uint64_t(len(b.In)),
		out: (*C.uint8_t)(unsafe.Pointer(&out[0])), out_size: C.uint64_t(len(out)),
		frames: ptrOrNil(b.frames), n_frames: C.uint32_t(n),
		blocks: ptrOrNil(b.blocks), n_blocks: C.uint32_t(len(b.blocks)),
		fse_tables: ptrOrNil(b.fseTables), n_fse_tables: C.uint32_t(len(b.fseTables)),
		fse_entries: ptrOrNil(b.fseEntries), n_fse_entries: C.uint32_t(len(b.fseEntries)),
		huf_tables: ptrOrNil(b.hufTables), n_huf_tables: C.uint32_t(len(b.hufTables)),
		huf_entries: ptrOrNil(b.hufEntries), n_huf_entries: C.uint32_t(len(b.hufEntries))}
	// cb holds Go pointers to pointer-free memory for the duration of this call only: allowed by the cgo rules
	// as long as cb itself is passed by pointer to C from Go-allocated, pinned-for-the-call memory.
	var pin runtime.Pinner
	pin.Pin(&in[0])
	pin.Pin(&out[0])
	if p := ptrOrNil(b.frames); p != nil {
		pin.Pin(p)
	}
	if p := ptrOrNil(b.blocks); p != nil {
		pin.Pin(p)
	}
	if p := ptrOrNil(b.fseTables); p != nil {
		pin.Pin(p)
	}
	if p := ptrOrNil(b.fseEntries); p != nil {
		pin.Pin(p)
	}
	if p := ptrOrNil(b.hufTables); p != nil {
		pin.Pin(p)
	}
	if p := ptrOrNil(b.hufEntries); p != nil {
		pin.Pin(p)
	}
	defer pin.Unpin()
	rc := C.mzd_decode_batch(x.c, &cb, (*C.int32_t)(unsafe.Pointer(&status[0])), (*C.uint64_t)(unsafe.Pointer(&outLen[0])))
	if rc >= C.MZD_ERR_DEVICE {
		return nil, nil, nil, errors.New(C.GoString(C.mzd_strerror(rc)) + ": " + x.lastError())
	}
	return out, status[:n], outLen[:n], nil // per-frame errors are in status[] (SentinelFor)
}

// BatchMark is a position in a Batch that Truncate can return to (every slice length and the output size).
type BatchMark struct {
	in, frames, blocks, fseTables, fseEntries, hufTables, hufEntries int
	outSize                                                         uint64
}

// Mark remembers the batch as it is now; Truncate(mark) drops everything appended since (a frame whose planning failed
// half-way: its frame slot, blocks, tables, cells and payload bytes), so that the slots of the frames planned afterwards
// stay consecutive and no unfinished frame reaches the device.
func (b *Batch) Mark() BatchMark {
	return BatchMark{len(b.In), len(b.frames), len(b.blocks), len(b.fseTables), len(b.fseEntries), len(b.hufTables), len(b.hufEntries), b.outSize}
}
func (b *Batch) Truncate(m BatchMark) {
	b.In, b.frames, b.blocks = b.In[:m.in], b.frames[:m.frames], b.blocks[:m.blocks]
	b.fseTables, b.fseEntries = b.fseTables[:m.fseTables], b.fseEntries[:m.fseEntries]
	b.hufTables, b.hufEntries = b.hufTables[:m.hufTables], b.hufEntries[:m.hufEntries]
	b.outSize = m.outSize
}

// OutOffset is where frame i's slab starts in the blob Decode returns.
func (b *Batch) OutOffset(i int) uint64 { return uint64(b.frames[i].out_offset) }
