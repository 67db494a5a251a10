// PlanNextBlock: the HOST-PLANNED route of north_star, written out against the reference's own types -- Go keeps frame /
// block / section header parsing and the table descriptions (structure/*.go, fse/fse.go), and instead of running the
// three hot loops it appends one descriptor per block to a gpu.Batch; Batch.Decode then makes ONE cgo call.
//
// UNVERIFIED (no Go toolchain in the build image) and NOT self-contained: it is the patch a sparkzstd maintainer applies
// INSIDE the reference's decompression package (it uses the reference's unexported fields), shown here as a file so that
// it can be reviewed and compiled there.  The two small changes it needs in the reference are marked NEEDS.
//
//go:build sparkzstd_patch

package decompression

import (
	"bufio"
	"io"

	"github.com/killingspark/sparkzstd/shim/go/gpu"
	"github.com/killingspark/sparkzstd/structure"
)

// gpuPlan is the per-frame planning state: which table indices the "previous block" carries
// (framedecompressor.go:283-294: Repeat mode / Treeless literals mean "the table used last", including an RLE one).
type gpuPlan struct {
	b                   *gpu.Batch
	bp                  *batchPlan
	prevHuf             uint32
	prevLL, prevOF, prevML uint32
}

// batchPlan is the per-BATCH planning state: the indices of the three predefined tables in THIS batch (added on first
// use; a package-level map would hand a second batch the first one's indices) -- one per gpu.Batch, used by one goroutine.
type batchPlan struct {
	b          *gpu.Batch
	predefined [3]uint32 // by kind (gpu.KindLL / KindOF / KindML); noTable until the batch has the table
}

func newBatchPlan(b *gpu.Batch) *batchPlan {
	return &batchPlan{b: b, predefined: [3]uint32{noTable, noTable, noTable}}
}

const noTable = 0xFFFFFFFF

// PlanFrame replaces decodeAllBlocks (framedecompressor.go:246-254) for one frame: after CheckMagicnum and
// DecodeFrameHeader it walks the blocks like DecodeNextBlock (:198-244) but records descriptors instead of decoding.
// It returns the index of the frame IN THE BATCH (what status / outLen / OutOffset are looked up with).  A frame that
// fails half-way is taken back out: the batch is truncated to the mark taken before BeginFrame, so that the frames after
// it keep consecutive slots and nothing unfinished reaches the device.
func (fd *FrameDecompressor) PlanFrame(bp *batchPlan) (int, error) {
	h := fd.frame.Header
	b := bp.b
	mark := b.Mark()
	predefinedBefore := bp.predefined
	slot := b.BeginFrame(uint64(h.WindowSize), uint64(h.FrameContentSize), h.FrameContentSize > 0 || h.Descriptor.GetSingleSegmentFlag())
	p := &gpuPlan{b: b, bp: bp, prevHuf: noTable, prevLL: noTable, prevOF: noTable, prevML: noTable}
	for !fd.CurrentBlock.Header.LastBlock {
		if err := fd.PlanNextBlock(p); err != nil {
			b.Truncate(mark) // frames, blocks, tables, cells and payload bytes this frame added
			bp.predefined = predefinedBefore
			return -1, err
		}
		fd.BlockCounter++
	}
	b.FinishFrame()
	return slot, nil
}

// PlanNextBlock is DecodeNextBlock (framedecompressor.go:198-244) up to and including the table descriptions.
func (fd *FrameDecompressor) PlanNextBlock(p *gpuPlan) error {
	if fd.CurrentBlock.Header.LastBlock {
		return ErrOutOfBlocks
	}
	if err := fd.DecodeNextBlockHeader(); err != nil { // :270-303 (3 header bytes, table carry-over of the CPU path)
		return err
	}
	size := int(fd.CurrentBlock.Header.BlockSize)
	switch fd.CurrentBlock.Header.Type {
	case structure.BlockTypeRaw: // :211-215
		buf := make([]byte, size)
		if _, err := io.ReadFull(fd.source, buf); err != nil {
			return err
		}
		p.b.AddRawBlock(buf)
		return nil
	case structure.BlockTypeCompressed:
		fd.limitedSource = &io.LimitedReader{R: fd.source, N: int64(size)}
		return fd.planCompressed(p, bufio.NewReader(fd.limitedSource), size)
	default: // RLE, :229-241
		v, err := fd.source.ReadByte()
		if err != nil {
			return err
		}
		p.b.AddRLEBlock(v, uint32(size))
		return nil
	}
}

// planCompressed is DecodeNextBlockContent (:93-126) with the stream decodes left out.
func (fd *FrameDecompressor) planCompressed(p *gpuPlan, src *bufio.Reader, blockSize int) error {
	cb := gpu.CompressedBlock{BlockSize: uint32(blockSize), HufTable: noTable, LLTable: noTable, OFTable: noTable, MLTable: noTable}
	ls := &fd.CurrentBlock.Literals

	// ---- literals section header (literals.go:209-244), Huffman tree description (huffman.go:40-107), jump table
	// NEEDS: LiteralSection.DecodeHeaderAndTree(source, prevBlock) = DecodeNextLiteralsSection (literals.go:209-289) cut
	// before its last step (the DecodeStream calls, :290-371): it leaves Header, TreeDesc.Weights / MaxBits and
	// CompressedData filled.
	if err := ls.DecodeHeaderAndTree(src, &fd.PreviousBlock); err != nil {
		return err
	}
	cb.LitRegen = uint32(ls.Header.RegeneratedSize)
	cb.LitPayload = ls.CompressedData
	switch ls.Header.Type {
	case structure.LiteralsBlockTypeRaw:
		cb.LitType = 0 // MZD_LIT_RAW
	case structure.LiteralsBlockTypeRLE:
		cb.LitType = 1 // MZD_LIT_RLE
	default: // Compressed / Treeless: both reach the device as MZD_LIT_HUF with the table resolved
		cb.LitType = 2
		cb.LitStreams = ls.Header.NumberOfStreams
		if ls.Header.Type == structure.LiteralsBlockTypeCompressed {
			// the device builds the decode table from the weights (k_huf_build = huffman.go:112-190)
			p.prevHuf = p.b.AddHuffmanWeights(ls.TreeDesc.Weights, ls.TreeDesc.MaxBits)
		} else if p.prevHuf == noTable {
			return structure.ErrNoHuffTableToCarryOver // literals.go:247-252
		}
		cb.HufTable = p.prevHuf
		if cb.LitStreams == 4 { // literals.go:313-361: three jump-table sizes + the computed fourth
			cb.LitStreamSize = [4]uint32{uint32(ls.Header.StreamSize1), uint32(ls.Header.StreamSize2), uint32(ls.Header.StreamSize3),
				uint32(ls.Header.CalcStreamsize4())}
		} else {
			cb.LitStreamSize[0] = uint32(ls.Header.CompressedSize)
		}
	}

	// ---- sequences section header, table descriptions (sequences.go:371-433, :275-370; fse.go:28-130)
	// NEEDS: SequencesSection.DecodeHeaderAndTableDescriptions(source, bytesLeft, prevBlock) = DecodeNextSequenceSection
	// (sequences.go:371-450) cut before DecodeSequences (:126-206), exposing per table its mode and, for Compressed mode,
	// the normalised counts ReadTabledescriptionFromBitstream (fse.go:28-130) parsed (Counts []int16, AccuracyLog).
	ss := &fd.CurrentBlock.Sequences
	used := ls.Header.CompressedSize + ls.Header.BytesUsedByHeader + ls.BytesUsedByTree
	if err := ss.DecodeHeaderAndTableDescriptions(src, blockSize-used, &fd.PreviousBlock); err != nil {
		return err
	}
	cb.NSeq = uint32(ss.Header.NumberOfSequences)
	if cb.NSeq == 0 && len(ss.Data) > 0 {
		// Zero sequences in the TWO-byte form (0x80 0x00): modes, tables and a bitstream follow all the same, and DecodeSequences
		// (sequences.go:126-208) reads the padding and the three initial states and wants the stream used up.  The device's sequence
		// stage has no chain to run for such a block: the verdict travels in mzd_block_desc.seq_status (ABI 9) and the execution
		// stage reports it in the stage's place.  Here the reference's own function gives it.
		if _, err := ss.DecodeSequences(); err != nil {
			cb.SeqStatus = gpu.StatusFor(err) // ErrBadPadding -> 8, ErrNotAllBitsUsed -> 11
		}
	}
	if cb.NSeq > 0 {
		cb.SeqPayload = ss.Data
		var err error
		if cb.LLTable, err = p.table(ss.Header.LiteralLengthsMode, ss.LLDescription, gpu.KindLL, &p.prevLL); err != nil {
			return err
		}
		if cb.OFTable, err = p.table(ss.Header.OffsetsMode, ss.OFDescription, gpu.KindOF, &p.prevOF); err != nil {
			return err
		}
		if cb.MLTable, err = p.table(ss.Header.MatchLengthsMode, ss.MLDescription, gpu.KindML, &p.prevML); err != nil {
			return err
		}
	}
	if fd.limitedSource.N != 0 { // framedecompressor.go:112-123
		return ErrCorruptSizes
	}
	p.b.AddCompressedBlock(&cb)
	return nil
}

// predefined tables are tables 0..2 of every batch in the device planner's layout; the Go side adds them once per
// batch on first use (predefined.go:22,52,70 as normalised counts) and keeps their indices in the batch's batchPlan.

// table resolves one of the three sequence tables to an index in the batch (sequences.go:275-370).
func (p *gpuPlan) table(mode structure.SymbolCompressionMode, d structure.TableDescription, kind int, prev *uint32) (uint32, error) {
	switch mode {
	case structure.SymbolCompressionModePredefined:
		idx := p.bp.predefined[kind]
		if idx == noTable {
			idx = p.b.AddFSECounts(d.AccuracyLog, d.Counts, kind) // the predefined distribution, as counts
			p.bp.predefined[kind] = idx
		}
		*prev = idx
	case structure.SymbolCompressionModeRLE:
		*prev = p.b.AddRLETable(d.RLESymbol, kind) // sequences.go:27-62
	case structure.SymbolCompressionModeFSECompressed:
		*prev = p.b.AddFSECounts(d.AccuracyLog, d.Counts, kind) // k_fse_build = fse.go:136-230
	default: // Repeat: the table used last (framedecompressor.go:283-294)
		if *prev == noTable {
			return 0, structure.ErrNoLLTableToCarryOver
		}
	}
	return *prev, nil
}

// DecodeFramesPlanned: DecodeFrames over the host-planned route.
func DecodeFramesPlanned(ctx *gpu.Context, frames []io.Reader) ([][]byte, []error, error) {
	var b gpu.Batch
	bp := newBatchPlan(&b)
	errs := make([]error, len(frames))
	slotOf := make([]int, len(frames)) // index in frames -> frame index in the batch (-1: not planned)
	for i, r := range frames {
		slotOf[i] = -1
		fd := NewFrameDecompressor(r, nil)
		if errs[i] = fd.CheckMagicnum(); errs[i] != nil {
			continue
		}
		if errs[i] = fd.DecodeFrameHeader(); errs[i] != nil {
			continue
		}
		slotOf[i], errs[i] = fd.PlanFrame(bp) // (a frame that fails half-way has been truncated out of the batch again)
	}
	out, status, outLen, err := b.Decode(ctx)
	if err != nil {
		return nil, nil, err
	}
	res := make([][]byte, len(frames))
	for i, k := range slotOf {
		if k < 0 {
			continue
		}
		if errs[i] = gpu.SentinelFor(status[k]); errs[i] == nil {
			o := b.OutOffset(k)
			res[i] = out[o : o+outLen[k]]
		}
	}
	return res, errs, nil
}
