{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE LambdaCase #-}
{-# LANGUAGE ScopedTypeVariables #-}

{-|
Module      : Codec.Compression.LZ4.Conduit.Batched
Description : Conduits that hand the LZ4 frame codec many blocks per call (for liblz4f_mi355x).

An addition to nh2/lz4-frame-conduit (SURVEY.md 8f, row N2).  The conduits of
"Codec.Compression.LZ4.Conduit" call @LZ4F_compressUpdate@ once per 16 KiB slice
(@Conduit.hsc:452-469@) and @LZ4F_decompress@ once per 16 KiB of output (@Conduit.hsc:638-669@) and
can only pass @lz4DefaultPreferences@ (@Conduit.hsc:340-347@).  Against @liblz4f_mi355x@ that is
correct but bound by PCIe round trips and kernel launches: one 64 KiB block per four calls.  The
conduits here give the library what it needs to be fast - caller-chosen preferences and hundreds
of blocks per call - through the same C ABI:

* 'compressWithPreferences' - the reference's slice loop with the caller's 'Preferences';
* 'compressBatched' - gathers @batchBytes@ of input in a page-locked buffer, then ONE
  @LZ4F_compressUpdate@ (a single call may create many blocks, @Conduit.hsc:326-333@);
* 'decompressBatched' / 'decompressBatchedWith' - walks the size words over the incoming chunks and hands runs of whole blocks
  to @lz4f_mi355x_fdec_blocks@ (bounded memory: one batch of input and the slabs in flight), which yields the output slab by slab.

The C++ mirror of exactly these three drivers is @lz4_frame_conduit_amd/csrc/conduit.cpp@; it is what
@tests/test_gpu_parity.py@ runs (@test_conduits_on_reference_test_inputs@, @test_batched_decoder_walks_whole_streams@).
THIS FILE HAS NOT BEEN COMPILED: the build image has no GHC.  It uses plain @foreign import ccall@
(no inline-c), so it needs nothing beyond what the package already depends on.
-}
module Codec.Compression.LZ4.Conduit.Batched
  ( compressWithPreferences
  , compressBatched
  , decompressBatched
  , decompressBatchedWith
  , defaultBatchBytes
  , defaultDecodeBatchBytes
  , appendBlockList
  ) where

import           Control.Concurrent (forkIO)
import           Control.Concurrent.MVar (MVar, newEmptyMVar, putMVar, takeMVar)
import           Control.Monad (unless, void, when)
import           Control.Monad.IO.Class (liftIO)
import           Control.Monad.IO.Unlift (MonadUnliftIO)
import           Control.Monad.Trans.Resource (MonadResource)
import           Data.Bits (shiftL, (.&.), (.|.))
import           Data.ByteString (ByteString, packCStringLen)
import qualified Data.ByteString as BS
import           Data.ByteString.Unsafe (unsafePackCString, unsafeUseAsCStringLen)
import           Data.Conduit
import           Data.IORef (newIORef, readIORef, writeIORef)
import           Data.Word (Word32)
import           Foreign.C.String (CString)
import           Foreign.C.Types (CChar, CSize (..), CUInt (..))
import           Foreign.Marshal.Alloc (alloca, allocaBytes, free, mallocBytes)
import           Foreign.Marshal.Utils (copyBytes, new)
import           Foreign.Ptr (FunPtr, Ptr, freeHaskellFunPtr, nullPtr, plusPtr)
import           Foreign.Storable (peek, peekByteOff)
import           UnliftIO.Exception (throwString)

import           Codec.Compression.LZ4.Conduit (BlockSizeID (..), FrameInfo (..), Preferences (..))
import           Codec.Compression.LZ4.CTypes (LZ4F_cctx)


-- * The C ABI (include/lz4f_mi355x.h).  Part 1 names are liblz4's; Part 2 are the library's own.

foreign import ccall unsafe "LZ4F_isError"                 c_isError :: CSize -> IO CUInt
foreign import ccall unsafe "LZ4F_getErrorName"            c_getErrorName :: CSize -> IO CString
foreign import ccall safe   "LZ4F_createCompressionContext" c_createCctx :: Ptr (Ptr LZ4F_cctx) -> CUInt -> IO CSize
foreign import ccall safe   "LZ4F_freeCompressionContext"  c_freeCctx :: Ptr LZ4F_cctx -> IO CSize
foreign import ccall safe   "LZ4F_compressBegin"           c_compressBegin :: Ptr LZ4F_cctx -> Ptr CChar -> CSize -> Ptr Preferences -> IO CSize
foreign import ccall unsafe "LZ4F_compressBound"           c_compressBound :: CSize -> Ptr Preferences -> IO CSize
foreign import ccall safe   "LZ4F_compressUpdate"          c_compressUpdate :: Ptr LZ4F_cctx -> Ptr CChar -> CSize -> Ptr CChar -> CSize -> Ptr () -> IO CSize
foreign import ccall safe   "LZ4F_compressEnd"             c_compressEnd :: Ptr LZ4F_cctx -> Ptr CChar -> CSize -> Ptr () -> IO CSize
-- page-locked host memory: the bulk path copies straight out of / into it (no staging copy)
foreign import ccall safe   "lz4f_mi355x_host_alloc"       c_hostAlloc :: CSize -> IO (Ptr CChar)
foreign import ccall safe   "lz4f_mi355x_host_free"        c_hostFree :: Ptr CChar -> IO ()
-- the output of a bulk decode is handed to a callback, slab by slab, in order
type YieldFn = Ptr () -> Ptr CChar -> CSize -> IO ()
foreign import ccall "wrapper" mkYieldFn :: YieldFn -> IO (FunPtr YieldFn)
-- a frame decoded batch by batch (include/lz4f_mi355x.h: lz4f_mi355x_fdec_*): header once, runs of whole blocks, the closing words
data Fdec
foreign import ccall unsafe "LZ4F_headerSize"              c_headerSize :: Ptr CChar -> CSize -> IO CSize
foreign import ccall safe   "lz4f_mi355x_fdec_create"      c_fdecCreate :: Ptr (Ptr Fdec) -> Ptr CChar -> CSize -> Ptr () -> IO CSize
foreign import ccall safe   "lz4f_mi355x_fdec_blocks"      c_fdecBlocks :: Ptr Fdec -> FunPtr YieldFn -> Ptr () -> Ptr CChar -> CSize -> IO CSize
foreign import ccall safe   "lz4f_mi355x_fdec_end"         c_fdecEnd :: Ptr Fdec -> Ptr CChar -> CSize -> IO CSize
foreign import ccall safe   "lz4f_mi355x_fdec_free"        c_fdecFree :: Ptr Fdec -> IO ()
-- the block list of a finished frame (include/lz4f_mi355x.h: host work only, no GPU)
foreign import ccall unsafe "lz4f_mi355x_blockListSize"     c_blockListSize :: Ptr CChar -> CSize -> IO CSize
foreign import ccall unsafe "lz4f_mi355x_appendBlockList"   c_appendBlockList :: Ptr CChar -> CSize -> CSize -> IO CSize


-- | Same message format as the reference's @handleLz4Error@ (@Conduit.hsc:149-160@).
checkLz4 :: IO CSize -> IO CSize
checkLz4 act = do
  r <- act
  isErr <- c_isError r
  when (isErr /= 0) $ do
    name <- unsafePackCString =<< c_getErrorName r
    throwString ("lz4frame error: " ++ show name)
  return r


-- | A finished LZ4 frame (this library's or any other encoder's, held whole) followed by its block list as a skippable
-- frame: every LZ4 reader skips it, @lz4f_mi355x_dev_decompressFrame@ finds the blocks through it instead of walking the
-- size words.  Throws the library's error (@ERROR_frameSize_wrong@ when the bytes are not exactly one frame).
appendBlockList :: ByteString -> IO ByteString
appendBlockList frame = unsafeUseAsCStringLen frame $ \(p, n) -> do
  extra <- checkLz4 (c_blockListSize p (fromIntegral n))
  if extra == 0 then return frame else do
    let total = n + fromIntegral extra
    allocaBytes total $ \buf -> do                      -- (released also when the call below throws)
      copyBytes buf p n
      r <- checkLz4 (c_appendBlockList buf (fromIntegral n) (fromIntegral total))
      packCStringLen (buf, fromIntegral r)


lz4fVersion :: CUInt
lz4fVersion = 100          -- LZ4F_VERSION

lz4fHeaderSizeMax :: CSize
lz4fHeaderSizeMax = 19     -- LZ4F_HEADER_SIZE_MAX

-- | 64 MiB: one slab of the library's host pipeline (csrc/pipeline.hip).
defaultBatchBytes :: Int
defaultBatchBytes = 64 * 1024 * 1024


data CompressState = CompressState
  { csCtx      :: !(Ptr LZ4F_cctx)
  , csPrefs    :: !(Ptr Preferences)
  , csIn       :: !(Ptr CChar)
  , csOut      :: !(Ptr CChar)
  , csOutSize  :: !CSize
  , csPinned   :: !Bool
  }


-- | The reference's 'compress' (16 KiB slices, one @LZ4F_compressUpdate@ each) with caller-chosen preferences.
compressWithPreferences :: (MonadUnliftIO m, MonadResource m) => Preferences -> ConduitT ByteString ByteString m ()
compressWithPreferences = compressGathering False (16 * 1024)


-- | Gathers @batchBytes@ of input (rounded up to whole blocks), then one @LZ4F_compressUpdate@: all completed blocks of the
-- batch are encoded by one pass of the GPU kernels.  The frame is an ordinary LZ4 frame; with linked blocks the context keeps
-- the 64 KiB of history across batches as liblz4's does.  Memory: the batch and its worst-case output, page-locked.
compressBatched :: (MonadUnliftIO m, MonadResource m) => Int -> Preferences -> ConduitT ByteString ByteString m ()
compressBatched batchBytes prefs = compressGathering True (roundUp (max 1 batchBytes)) prefs
  where
    blockBytes = case blockSizeID (frameInfo prefs) of
      LZ4F_max256KB -> 256 * 1024
      LZ4F_max1MB   -> 1024 * 1024
      LZ4F_max4MB   -> 4 * 1024 * 1024
      _             -> 64 * 1024
    roundUp n = ((n + blockBytes - 1) `div` blockBytes) * blockBytes


compressGathering :: forall m . (MonadUnliftIO m, MonadResource m) => Bool -> Int -> Preferences -> ConduitT ByteString ByteString m ()
compressGathering pinned batch prefs = bracketP acquire release run
  where
    allocBuf :: CSize -> IO (Ptr CChar)
    allocBuf n
      | pinned = do
          p <- c_hostAlloc n
          when (p == nullPtr) $ throwString "lz4frame error: \"ERROR_allocation_failed\""
          return p
      | otherwise = mallocBytes (fromIntegral n)

    acquire :: IO CompressState
    acquire = do
      ctx <- alloca $ \pp -> do
        _ <- checkLz4 (c_createCctx pp lz4fVersion)
        peek pp
      prefsPtr <- new prefs
      outSize <- checkLz4 (c_compressBound (fromIntegral batch + lz4fHeaderSizeMax) prefsPtr)
      inBuf <- allocBuf (fromIntegral batch)
      outBuf <- allocBuf outSize
      return (CompressState ctx prefsPtr inBuf outBuf outSize pinned)

    release :: CompressState -> IO ()
    release s = do
      void (c_freeCctx (csCtx s))
      free (csPrefs s)
      if csPinned s then c_hostFree (csIn s) >> c_hostFree (csOut s) else free (csIn s) >> free (csOut s)

    run :: CompressState -> ConduitT ByteString ByteString m ()
    run s = do
      let emit n = when (n > 0) $ yield =<< liftIO (packCStringLen (csOut s, fromIntegral n))
      -- header (Conduit.hsc:292-296)
      emit =<< liftIO (checkLz4 (c_compressBegin (csCtx s) (csOut s) (csOutSize s) (csPrefs s)))
      -- batches
      let flush used = when (used > 0) $
            emit =<< liftIO (checkLz4 (c_compressUpdate (csCtx s) (csOut s) (csOutSize s) (csIn s) (fromIntegral used) nullPtr))
          gather used = await >>= \case
            Nothing -> flush used
            Just bs -> place used bs
          place used bs
            | BS.null bs = gather used
            | otherwise = do
                let (now, later) = BS.splitAt (batch - used) bs
                liftIO $ unsafeUseAsCStringLen now $ \(p, l) -> copyBytes (csIn s `plusPtr` used) p l
                let used' = used + BS.length now
                if used' == batch then flush used' >> place 0 later else place used' later
      gather 0
      -- EndMark and content checksum (Conduit.hsc:318-324); srcSize 0 is the bound for LZ4F_compressEnd
      emit =<< liftIO (checkLz4 (c_compressEnd (csCtx s) (csOut s) (csOutSize s) nullPtr))


data Handoff = Chunk !ByteString | Done !CSize      -- a piece of output; the C call's result


-- | 256 MiB of whole blocks per @lz4f_mi355x_fdec_blocks@ call: four slabs of the library's host pipeline in flight.
defaultDecodeBatchBytes :: Int
defaultDecodeBatchBytes = 256 * 1024 * 1024


-- | 'decompressBatchedWith' 'defaultDecodeBatchBytes'.
decompressBatched :: forall m . (MonadUnliftIO m, MonadResource m) => ConduitT ByteString ByteString m ()
decompressBatched = decompressBatchedWith defaultDecodeBatchBytes


-- | Decodes a stream of LZ4 frames through the bulk path in BOUNDED memory.  The conduit walks the size words over the chunks as
-- they arrive (a 4-byte read per block), cuts runs of whole blocks - @batchBytes@ of them at a time - and hands each run to
-- @lz4f_mi355x_fdec_blocks@, which keeps slabs of blocks in flight on the GPU(s) and hands the output back slab by slab, in
-- order.  What is held: one batch of input, the chunk being cut, the slabs in flight - whatever the stream's length (the
-- reference's 'decompress' holds one @max(hint, 16 KiB)@ buffer, @Conduit.hsc:634-659@; an earlier version of this conduit gathered
-- the whole stream first).  Linked frames carry their 64 KiB of history from run to run inside the decoder object; content
-- checksum and contentSize are verified over all runs by @lz4f_mi355x_fdec_end@.
-- Unlike the reference's 'decompress' (which stops at the first EndMark and cannot read headers with a dictID) this decodes every
-- frame of the stream, skips skippable frames and accepts every header the format allows - what calling @LZ4F_decompress@ in a
-- loop does.  Build with @-threaded@: the C call runs beside the conduit and calls back into Haskell.
decompressBatchedWith :: forall m . (MonadUnliftIO m, MonadResource m) => Int -> ConduitT ByteString ByteString m ()
decompressBatchedWith batchBytes0 = go BS.empty 0 False
  where
    batchBytes = max (1024 * 1024) batchBytes0

    le32 :: ByteString -> Int -> Word32
    le32 b i = fromIntegral (BS.index b i) .|. (fromIntegral (BS.index b (i + 1)) `shiftL` 8)
           .|. (fromIntegral (BS.index b (i + 2)) `shiftL` 16) .|. (fromIntegral (BS.index b (i + 3)) `shiftL` 24)

    -- at least n bytes of input in hand (Nothing: the stream ended first); what came in addition is counted
    ensure :: Int -> ByteString -> Int -> ConduitT ByteString ByteString m (Maybe ByteString, Int)
    ensure n have seen
      | BS.length have >= n = return (Just have, seen)
      | otherwise = await >>= \case
          Nothing -> return (Nothing, seen)
          Just bs -> ensure n (if BS.null have then bs else have <> bs) (seen + BS.length bs)

    endedEarly :: ConduitT ByteString ByteString m a
    endedEarly = throwString "lz4 decompress error: stream ended before EndMark"

    -- between frames: pending input, bytes seen so far, whether any frame was seen
    go :: ByteString -> Int -> Bool -> ConduitT ByteString ByteString m ()
    go pend seen anyFrame = do
      (m1, seen1) <- ensure 1 pend seen
      case m1 of
        Nothing
          | anyFrame  -> return ()
          | otherwise -> throwString ("lz4 decompress error: not enough bytes for header; expected 5, got " ++ show seen1)
        Just p1 -> do
          (m4, seen4) <- ensure 4 p1 seen1
          case m4 of
            Nothing
              | not anyFrame && seen4 < 5 -> throwString ("lz4 decompress error: not enough bytes for header; expected 5, got " ++ show seen4)
              | otherwise -> throwString "lz4frame error: \"ERROR_frameHeader_incomplete\""
            Just p4
              | le32 p4 0 .&. 0xFFFFFFF0 == 0x184D2A50 -> do              -- skippable frame: dropped as it arrives
                  (m8, seen8) <- ensure 8 p4 seen4
                  p8 <- maybe (throwString "lz4frame error: \"ERROR_frameHeader_incomplete\"") return m8
                  (rest, seen') <- skipBytes (fromIntegral (le32 p8 4)) (BS.drop 8 p8) seen8
                  go rest seen' True
              | otherwise -> do
                  (rest, seen') <- oneFrame p4 seen4
                  go rest seen' True

    skipBytes :: Int -> ByteString -> Int -> ConduitT ByteString ByteString m (ByteString, Int)
    skipBytes n have seen
      | BS.length have >= n = return (BS.drop n have, seen)
      | otherwise = await >>= \case
          Nothing -> endedEarly
          Just bs -> skipBytes (n - BS.length have) bs (seen + BS.length bs)

    -- One frame: header -> decoder object; runs of whole blocks; the closing words.
    oneFrame :: ByteString -> Int -> ConduitT ByteString ByteString m (ByteString, Int)
    oneFrame p0 seen0 = do
      (m7, seen7) <- ensure 7 p0 seen0
      p7 <- case m7 of
        Just x -> return x
        Nothing | seen7 < 5 -> throwString ("lz4 decompress error: not enough bytes for header; expected 5, got " ++ show seen7)
                | otherwise -> throwString "lz4frame error: \"ERROR_frameHeader_incomplete\""
      hs <- liftIO $ unsafeUseAsCStringLen p7 $ \(p, n) -> checkLz4 (c_headerSize p (fromIntegral n))
      (mh, seenh) <- ensure (fromIntegral hs) p7 seen7
      ph <- maybe (throwString "lz4frame error: \"ERROR_frameHeader_incomplete\"") return mh
      bracketP (openDec (BS.take (fromIntegral hs) ph)) (\(d, _, _, _) -> c_fdecFree d) $ \(dec, bck, cck, maxBlock) -> do
        let crc = if bck then 4 else 0
            tailLen = if cck then 8 else 4
            -- `run` bytes of whole blocks lie at the front of `have`
            blocks have seen run = do
              (mw, seenw) <- ensure (run + 4) have seen
              hw <- maybe endedEarly return mw
              let w = le32 hw run
                  csz = fromIntegral (w .&. 0x7FFFFFFF) :: Int
              if w == 0
                then do
                  when (run > 0) $ runBlocks dec (BS.take run hw)
                  (mt, seent) <- ensure (run + tailLen) hw seenw
                  ht <- maybe endedEarly return mt
                  _ <- liftIO $ unsafeUseAsCStringLen (BS.take tailLen (BS.drop run ht)) $ \(p, n) -> checkLz4 (c_fdecEnd dec p (fromIntegral n))
                  return (BS.drop (run + tailLen) ht, seent)
                else do
                  when (csz > maxBlock) $ throwString "lz4frame error: \"ERROR_maxBlockSize_invalid\""
                  let step = 4 + csz + crc
                  (mb, seenb) <- ensure (run + step) hw seenw
                  hb <- maybe endedEarly return mb
                  if run + step >= batchBytes
                    then runBlocks dec (BS.take (run + step) hb) >> blocks (BS.drop (run + step) hb) seenb 0
                    else blocks hb seenb (run + step)
        blocks (BS.drop (fromIntegral hs) ph) seenh 0

    openDec :: ByteString -> IO (Ptr Fdec, Bool, Bool, Int)
    openDec hdr = unsafeUseAsCStringLen hdr $ \(p, n) -> alloca $ \pp -> allocaBytes 32 $ \fi -> do
      _ <- checkLz4 (c_fdecCreate pp p (fromIntegral n) fi)
      d <- peek pp
      -- LZ4F_frameInfo_t (CTypes.hsc:155-232): blockSizeID u32 @0, contentChecksumFlag u32 @8, blockChecksumFlag u32 @28
      bsid <- peekByteOff fi 0 :: IO Word32
      cckF <- peekByteOff fi 8 :: IO Word32
      bckF <- peekByteOff fi 28 :: IO Word32
      let maxBlock = case bsid of { 5 -> 256 * 1024; 6 -> 1024 * 1024; 7 -> 4 * 1024 * 1024; _ -> 64 * 1024 }
      return (d, bckF /= 0, cckF /= 0, maxBlock)

    -- One run of whole blocks.  The C call runs on its own thread and hands every piece of output over through a one-place MVar,
    -- so the conduit yields while the next slab is being decoded and nothing piles up.  If the conduit is abandoned downstream,
    -- the cancel flag makes the callback drop what is left, and the call runs to its end without blocking.
    runBlocks :: Ptr Fdec -> ByteString -> ConduitT ByteString ByteString m ()
    runBlocks dec run = bracketP setup teardown $ \(box, _, finished, _) ->
      let loop = liftIO (takeMVar box) >>= \case
            Chunk bs -> yield bs >> loop
            Done r -> do
              liftIO (writeIORef finished True)
              void (liftIO (checkLz4 (return r)))                           -- same exception text as the reference's
      in loop
      where
        setup = do
          box <- newEmptyMVar :: IO (MVar Handoff)
          cancelled <- newIORef False
          finished <- newIORef False
          cb <- mkYieldFn $ \_ dat n -> do
            gone <- readIORef cancelled
            unless gone $ putMVar box . Chunk =<< packCStringLen (dat, fromIntegral n)
          _ <- forkIO $ unsafeUseAsCStringLen run $ \(p, n) -> do
            r <- c_fdecBlocks dec cb nullPtr p (fromIntegral n)
            putMVar box (Done r)
          return (box, cancelled, finished, cb)
        teardown (box, cancelled, finished, cb) = do
          writeIORef cancelled True
          -- abandoned before the end: take what the callback is still handing over until the call says it is done;
          -- only then may the callback be freed
          let drain = takeMVar box >>= \case
                Done _  -> return ()
                Chunk _ -> drain
          seen <- readIORef finished
          unless seen drain
          freeHaskellFunPtr cb
